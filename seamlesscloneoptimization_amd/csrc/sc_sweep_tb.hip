// sc_sweep_tb.hip -- temporally blocked sweep kernels for gfx950: T full sweeps of the 5-point
// smoother per launch, field held in REGISTERS, row hand-off between waves through LDS.
//
// Geometry.  One wave64 spans a 256-float row segment (64 lanes x float4, 16-B aligned global
// loads); a wave owns R consecutive rows, a workgroup NW waves: a 256 x (NW*R) region.  Each
// lane keeps its 4 x R values of U and of the RHS in VGPRs for the whole launch, so per sweep
// only two things move: the left/right neighbour of a lane's float4 (one wave shuffle) and
// the top/bottom row of each wave's band (one ds_write_b128 + one ds_read_b128 per lane, double
// buffered so a single s_barrier per step suffices).  The region overlaps its neighbours by a
// halo of HX = 4 columns and HY rows (HY = 2T red-black, T Jacobi): values at depth d from the
// region edge stay exact for d steps, so the inner (256-8) x (NW*R-2HY) tile is exact after the
// launch and is the only part written back.  Results are bit-identical to T global sweeps.
//
// HBM traffic per launch ~ (4 B + 4 B)/efficiency + 4 B per unknown for T sweeps, against
// 12 B x T algorithmic -- hence "effective" bandwidth above the HBM roof for T > 1.
#include "sc_common.h"

namespace sc {

constexpr int TB_HX = 4; // column halo (one float4)

template <int R>
__device__ __forceinline__ void tb_load(const float *__restrict__ p, int P, int H, int x, int y0, float4 (&v)[R])
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < P) v[r] = *reinterpret_cast<const float4 *>(p + (size_t)y * P + x);
    }
}

// ---------------------------------------------------------------------------- red-black
template <int T, int NW, int R, bool SOR>
__global__ __launch_bounds__(NW * 64) void k_rb_tb(Field Uin, Field Uout, Field F, float omega)
{
    constexpr int HY = 2 * T, RH = NW * R;
    static_assert(2 * T <= TB_HX, "column halo too small for this depth");
    __shared__ float4 edge[2][NW][2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int W = Uin.W, H = Uin.H, P = Uin.pitch;
    const int x = blockIdx.x * (256 - 2 * TB_HX) - TB_HX + 4 * lane;
    const int ry = blockIdx.y * (RH - 2 * HY) - HY;
    const int y0 = ry + wv * R;
    float4 u[R], f[R];
    tb_load<R>(Uin.at(c), P, H, x, y0, u);
    tb_load<R>(F.at(c), P, H, x, y0, f);
    const bool x0ok = (x + 0 >= 1) && (x + 0 <= W - 2), x1ok = (x + 1 >= 1) && (x + 1 <= W - 2);
    const bool x2ok = (x + 2 >= 1) && (x + 2 <= W - 2), x3ok = (x + 3 >= 1) && (x + 3 <= W - 2);
    edge[0][wv][0][lane] = u[0];
    edge[0][wv][1][lane] = u[R - 1];
    __syncthreads();
#pragma unroll
    for (int step = 0; step < 2 * T; ++step) {
        const int buf = step & 1, color = step & 1;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 up = (wv > 0) ? edge[buf][wv - 1][1][lane] : zero;
        const float4 dn = (wv < NW - 1) ? edge[buf][wv + 1][0][lane] : zero;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + r;
            const bool yok = (y >= 1) && (y <= H - 2);
            const float4 a = (r == 0) ? up : u[r - 1];
            const float4 b = (r == R - 1) ? dn : u[r + 1];
            float4 cur = u[r];
            // x is a multiple of 4, so the colour of component k depends on (y + k) only: the
            // branch below is wave-uniform
            if (((y + color) & 1) == 0) {
                float l = __shfl_up(cur.w, 1, 64);
                if (lane == 0) l = 0.f;
                const float g0 = 0.25f * (((l + cur.y) + (a.x + b.x)) - f[r].x);
                const float g2 = 0.25f * (((cur.y + cur.w) + (a.z + b.z)) - f[r].z);
                const float n0 = SOR ? (cur.x + omega * (g0 - cur.x)) : g0;
                const float n2 = SOR ? (cur.z + omega * (g2 - cur.z)) : g2;
                if (yok && x0ok) cur.x = n0;
                if (yok && x2ok) cur.z = n2;
            } else {
                float rr = __shfl_down(cur.x, 1, 64);
                if (lane == 63) rr = 0.f;
                const float g1 = 0.25f * (((cur.x + cur.z) + (a.y + b.y)) - f[r].y);
                const float g3 = 0.25f * (((cur.z + rr) + (a.w + b.w)) - f[r].w);
                const float n1 = SOR ? (cur.y + omega * (g1 - cur.y)) : g1;
                const float n3 = SOR ? (cur.w + omega * (g3 - cur.w)) : g3;
                if (yok && x1ok) cur.y = n1;
                if (yok && x3ok) cur.w = n3;
            }
            u[r] = cur;
        }
        if (step + 1 < 2 * T) {
            edge[buf ^ 1][wv][0][lane] = u[0];
            edge[buf ^ 1][wv][1][lane] = u[R - 1];
            __syncthreads();
        }
    }
    // write back the exact inner tile (ring rows/columns are copied through unchanged)
    if (lane == 0 || lane == 63 || x >= P || x >= W) return;
    float *__restrict__ out = Uout.at(c);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yr = wv * R + r, y = y0 + r;
        if (yr >= HY && yr < RH - HY && y >= 0 && y < H) *reinterpret_cast<float4 *>(out + (size_t)y * P + x) = u[r];
    }
}

// ---------------------------------------------------------------------------- Jacobi
template <int T, int NW, int R>
__global__ __launch_bounds__(NW * 64) void k_jacobi_tb(Field Uin, Field Uout, Field F)
{
    constexpr int HY = T, RH = NW * R;
    static_assert(T <= TB_HX, "column halo too small for this depth");
    __shared__ float4 edge[2][NW][2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int W = Uin.W, H = Uin.H, P = Uin.pitch;
    const int x = blockIdx.x * (256 - 2 * TB_HX) - TB_HX + 4 * lane;
    const int ry = blockIdx.y * (RH - 2 * HY) - HY;
    const int y0 = ry + wv * R;
    float4 u[R], f[R];
    tb_load<R>(Uin.at(c), P, H, x, y0, u);
    tb_load<R>(F.at(c), P, H, x, y0, f);
    const bool x0ok = (x + 0 >= 1) && (x + 0 <= W - 2), x1ok = (x + 1 >= 1) && (x + 1 <= W - 2);
    const bool x2ok = (x + 2 >= 1) && (x + 2 <= W - 2), x3ok = (x + 3 >= 1) && (x + 3 <= W - 2);
    edge[0][wv][0][lane] = u[0];
    edge[0][wv][1][lane] = u[R - 1];
    __syncthreads();
#pragma unroll
    for (int step = 0; step < T; ++step) {
        const int buf = step & 1;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 prev = (wv > 0) ? edge[buf][wv - 1][1][lane] : zero;     // old row above
        const float4 dn = (wv < NW - 1) ? edge[buf][wv + 1][0][lane] : zero;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + r;
            const bool yok = (y >= 1) && (y <= H - 2);
            const float4 cur = u[r];
            const float4 b = (r == R - 1) ? dn : u[r + 1];
            float l = __shfl_up(cur.w, 1, 64), rr = __shfl_down(cur.x, 1, 64);
            if (lane == 0) l = 0.f;
            if (lane == 63) rr = 0.f;
            float4 nw = cur;
            const float n0 = 0.25f * (((l + cur.y) + (prev.x + b.x)) - f[r].x);
            const float n1 = 0.25f * (((cur.x + cur.z) + (prev.y + b.y)) - f[r].y);
            const float n2 = 0.25f * (((cur.y + cur.w) + (prev.z + b.z)) - f[r].z);
            const float n3 = 0.25f * (((cur.z + rr) + (prev.w + b.w)) - f[r].w);
            if (yok && x0ok) nw.x = n0;
            if (yok && x1ok) nw.y = n1;
            if (yok && x2ok) nw.z = n2;
            if (yok && x3ok) nw.w = n3;
            prev = cur;
            u[r] = nw;
        }
        if (step + 1 < T) {
            edge[buf ^ 1][wv][0][lane] = u[0];
            edge[buf ^ 1][wv][1][lane] = u[R - 1];
            __syncthreads();
        }
    }
    if (lane == 0 || lane == 63 || x >= P || x >= W) return;
    float *__restrict__ out = Uout.at(c);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yr = wv * R + r, y = y0 + r;
        if (yr >= HY && yr < RH - HY && y >= 0 && y < H) *reinterpret_cast<float4 *>(out + (size_t)y * P + x) = u[r];
    }
}

// ---------------------------------------------------------------------------- launchers
constexpr int TB_NW = 8, TB_R = 8;

template <int T, bool SOR>
static void launch_rb_t(Field Uin, Field Uout, Field F, float omega, hipStream_t s)
{
    constexpr int RH = TB_NW * TB_R, HY = 2 * T;
    dim3 grid((Uin.W + (256 - 2 * TB_HX) - 1) / (256 - 2 * TB_HX), (Uin.H + (RH - 2 * HY) - 1) / (RH - 2 * HY), Uin.C);
    hipLaunchKernelGGL((k_rb_tb<T, TB_NW, TB_R, SOR>), grid, dim3(TB_NW * 64), 0, s, Uin, Uout, F, omega);
}

bool launch_rb_tb(Field Uin, Field Uout, Field F, int sweeps, float omega, hipStream_t s)
{
    const bool sor = omega != 1.0f;
    switch (sweeps) {
    case 1: sor ? launch_rb_t<1, true>(Uin, Uout, F, omega, s) : launch_rb_t<1, false>(Uin, Uout, F, omega, s); return true;
    case 2: sor ? launch_rb_t<2, true>(Uin, Uout, F, omega, s) : launch_rb_t<2, false>(Uin, Uout, F, omega, s); return true;
    default: return false;
    }
}

template <int T>
static void launch_jacobi_t(Field Uin, Field Uout, Field F, hipStream_t s)
{
    constexpr int RH = TB_NW * TB_R, HY = T;
    dim3 grid((Uin.W + (256 - 2 * TB_HX) - 1) / (256 - 2 * TB_HX), (Uin.H + (RH - 2 * HY) - 1) / (RH - 2 * HY), Uin.C);
    hipLaunchKernelGGL((k_jacobi_tb<T, TB_NW, TB_R>), grid, dim3(TB_NW * 64), 0, s, Uin, Uout, F);
}

bool launch_jacobi_tb(Field Uin, Field Uout, Field F, int sweeps, hipStream_t s)
{
    switch (sweeps) {
    case 1: launch_jacobi_t<1>(Uin, Uout, F, s); return true;
    case 2: launch_jacobi_t<2>(Uin, Uout, F, s); return true;
    case 3: launch_jacobi_t<3>(Uin, Uout, F, s); return true;
    case 4: launch_jacobi_t<4>(Uin, Uout, F, s); return true;
    default: return false;
    }
}

int tb_max_depth(int method) { return method == SC_METHOD_JACOBI ? 4 : 2; }

} // namespace sc
