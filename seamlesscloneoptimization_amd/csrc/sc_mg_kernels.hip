// sc_mg_kernels.hip -- multigrid kernels for gfx950: general-coefficient red-black smoother
// for the coarse levels, residual field (double arithmetic), normalised-transpose restriction
// and bilinear prolongation with correction.  See MGDim in sc_common.h for the level geometry
// (uniform spacing except the last interval in each direction).
//
// No reference counterpart: the reference solves the Poisson system directly with a DST
// (seamlessClone_imp.cpp:1814-1896).  The multigrid converges to that system's solution for
// any ROI size; its components are checked against oracle/mg_np.py.
#include "sc_common.h"

namespace sc {

// ---- general red-black half sweep (levels >= 1; ring = 0) ---------------------------------
template <bool SOR>
__global__ __launch_bounds__(256) void k_rb_half_gen(Field U, Field F, int color, float omega, MGGeom g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int W = U.W, H = U.H, P = U.pitch;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int y = blockIdx.y * 4 + wv;
    if (y < 1 || y > H - 2 || x >= P) return;
    float *__restrict__ row = U.at(c) + (size_t)y * P;
    const float4 c4 = *reinterpret_cast<const float4 *>(row + x);
    const float4 u4 = *reinterpret_cast<const float4 *>(row - P + x);
    const float4 d4 = *reinterpret_cast<const float4 *>(row + P + x);
    const float4 f4 = *reinterpret_cast<const float4 *>(F.at(c) + (size_t)y * P + x);
    float l = __shfl_up(c4.w, 1, 64), r = __shfl_down(c4.x, 1, 64);
    if (lane == 0) l = (x > 0) ? row[x - 1] : 0.f;
    if (lane == 63) r = (x + 4 < P) ? row[x + 4] : 0.f;
    const float cn = (y == g.y.n) ? g.y.cw_last : 1.0f;
    const float dy = (y == g.y.n) ? g.y.d_last : 2.0f;
    float4 o = c4;
    const int par = (x + y + color) & 1;
#define SC_GEN_UPD(dst, XI, L, R, UU, DD, FF)                                     \
    if ((XI) >= 1 && (XI) <= W - 2) {                                             \
        const float cw = ((XI) == g.x.n) ? g.x.cw_last : 1.0f;                    \
        const float dx = ((XI) == g.x.n) ? g.x.d_last : 2.0f;                     \
        const float gs = (((cw * (L) + (R)) + (cn * (UU) + (DD))) - (FF)) / (dx + dy); \
        dst = SOR ? (dst + omega * (gs - dst)) : gs;                              \
    }
    if (par == 0) {
        SC_GEN_UPD(o.x, x + 0, l, c4.y, u4.x, d4.x, f4.x)
        SC_GEN_UPD(o.z, x + 2, c4.y, c4.w, u4.z, d4.z, f4.z)
    } else {
        SC_GEN_UPD(o.y, x + 1, c4.x, c4.z, u4.y, d4.y, f4.y)
        SC_GEN_UPD(o.w, x + 3, c4.z, r, u4.w, d4.w, f4.w)
    }
#undef SC_GEN_UPD
    *reinterpret_cast<float4 *>(row + x) = o;
}

void launch_rb_half_gen(Field U, Field F, int color, float omega, MGGeom g, hipStream_t s)
{
    dim3 grid((U.W + 255) / 256, (U.H + 3) / 4, U.C);
    if (omega == 1.0f)
        hipLaunchKernelGGL(k_rb_half_gen<false>, grid, dim3(256), 0, s, U, F, color, omega, g);
    else
        hipLaunchKernelGGL(k_rb_half_gen<true>, grid, dim3(256), 0, s, U, F, color, omega, g);
}

// ---- residual field R = F - A U, evaluated in double from the float32 values ---------------
// (float32 evaluation cancels catastrophically on smooth error: at 2048^2 it leaves ~1 grey
// level of smooth error invisible; the double evaluation is exact for float inputs.)
__global__ __launch_bounds__(256) void k_residual_field(Field U, Field F, Field R, MGGeom g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int W = U.W, H = U.H, P = U.pitch;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int y = blockIdx.y * 4 + wv;
    if (y < 1 || y > H - 2 || x >= P) return;
    const float *__restrict__ row = U.at(c) + (size_t)y * P;
    const float4 c4 = *reinterpret_cast<const float4 *>(row + x);
    const float4 u4 = *reinterpret_cast<const float4 *>(row - P + x);
    const float4 d4 = *reinterpret_cast<const float4 *>(row + P + x);
    const float4 f4 = *reinterpret_cast<const float4 *>(F.at(c) + (size_t)y * P + x);
    float l = __shfl_up(c4.w, 1, 64), r = __shfl_down(c4.x, 1, 64);
    if (lane == 0) l = (x > 0) ? row[x - 1] : 0.f;
    if (lane == 63) r = (x + 4 < P) ? row[x + 4] : 0.f;
    const double cn = (y == g.y.n) ? (double)g.y.cw_last : 1.0;
    const double dy = (y == g.y.n) ? (double)g.y.d_last : 2.0;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#define SC_RESF(dst, XI, CC, L, R_, UU, DD, FF)                                                     \
    if ((XI) >= 1 && (XI) <= W - 2) {                                                               \
        const double cw = ((XI) == g.x.n) ? (double)g.x.cw_last : 1.0;                              \
        const double dx = ((XI) == g.x.n) ? (double)g.x.d_last : 2.0;                               \
        const double s = ((cw * (double)(L) + (double)(R_)) + (cn * (double)(UU) + (double)(DD))) - \
                         (dx + dy) * (double)(CC);                                                  \
        dst = (float)((double)(FF) - s);                                                            \
    }
    SC_RESF(o.x, x + 0, c4.x, l, c4.y, u4.x, d4.x, f4.x)
    SC_RESF(o.y, x + 1, c4.y, c4.x, c4.z, u4.y, d4.y, f4.y)
    SC_RESF(o.z, x + 2, c4.z, c4.y, c4.w, u4.z, d4.z, f4.z)
    SC_RESF(o.w, x + 3, c4.w, c4.z, r, u4.w, d4.w, f4.w)
#undef SC_RESF
    *reinterpret_cast<float4 *>(R.at(c) + (size_t)y * P + x) = o;
}

void launch_residual_field(Field U, Field F, Field R, MGGeom g, hipStream_t s)
{
    dim3 grid((U.W + 255) / 256, (U.H + 3) / 4, U.C);
    hipLaunchKernelGGL(k_residual_field, grid, dim3(256), 0, s, U, F, R, g);
}

// ---- restriction: Fc = 4 * (row-normalised transpose of the interpolation) applied to R ----
// Coarse point I gathers fine points 2I-1, 2I, 2I+1 with weights 1/2, 1, 1/2; the last coarse
// point takes the (up to two) tail points with their interpolation weights instead.
__device__ __forceinline__ void restrict_weights(const MGDim &d, int I, float w[4], float &inv)
{
    w[0] = 0.5f; w[1] = 1.0f;
    if (I < d.nc) { w[2] = 0.5f; w[3] = 0.0f; inv = 0.5f; }
    else          { w[2] = d.tw1; w[3] = d.tw2; inv = d.inv_last; }
}

__global__ __launch_bounds__(256) void k_restrict(Field R, Field Fc, MGGeom g)
{
    const int I = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
    const int J = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
    const int c = blockIdx.z;
    if (I > g.x.nc || J > g.y.nc) return;
    float wx[4], wy[4], ix, iy;
    restrict_weights(g.x, I, wx, ix);
    restrict_weights(g.y, J, wy, iy);
    const float *__restrict__ r = R.at(c);
    const int P = R.pitch;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int y = 2 * J - 1 + a;
        if (wy[a] == 0.f || y > g.y.n) continue;
        float rowacc = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = 2 * I - 1 + b;
            if (wx[b] == 0.f || x > g.x.n) continue;
            rowacc += wx[b] * r[(size_t)y * P + x];
        }
        acc += wy[a] * rowacc;
    }
    Fc.at(c)[(size_t)J * Fc.pitch + I] = 4.0f * (acc * (ix * iy));
}

void launch_restrict(Field R, Field Fc, MGGeom g, hipStream_t s)
{
    dim3 grid((g.x.nc + 63) / 64, (g.y.nc + 3) / 4, R.C);
    hipLaunchKernelGGL(k_restrict, grid, dim3(256), 0, s, R, Fc, g);
}

// ---- prolongation + correction: Uf += P Uc on the fine interior ------------------------------
__device__ __forceinline__ void interp_1d(const MGDim &d, int i, int &I0, int &I1, float &w0, float &w1)
{
    if (i <= 2 * d.nc) {
        if ((i & 1) == 0) { I0 = i >> 1; I1 = I0; w0 = 1.0f; w1 = 0.0f; }
        else { I0 = (i - 1) >> 1; I1 = I0 + 1; w0 = 0.5f; w1 = 0.5f; }
    } else {
        I0 = d.nc; I1 = d.nc; w1 = 0.0f;
        w0 = (i - 2 * d.nc == 1) ? d.tw1 : d.tw2;
    }
}

__global__ __launch_bounds__(256) void k_prolong_add(Field Uc, Field Uf, MGGeom g, unsigned *__restrict__ maxcorr)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
    const int c = blockIdx.z;
    float corr = 0.f;
    if (x <= g.x.n && y <= g.y.n) {
        int I0, I1, J0, J1;
        float wx0, wx1, wy0, wy1;
        interp_1d(g.x, x, I0, I1, wx0, wx1);
        interp_1d(g.y, y, J0, J1, wy0, wy1);
        const float *__restrict__ e = Uc.at(c);
        const int Pc = Uc.pitch;
        const float top = wx0 * e[(size_t)J0 * Pc + I0] + wx1 * e[(size_t)J0 * Pc + I1];
        const float bot = wx0 * e[(size_t)J1 * Pc + I0] + wx1 * e[(size_t)J1 * Pc + I1];
        corr = wy0 * top + wy1 * bot;
        float *u = Uf.at(c) + (size_t)y * Uf.pitch + x;
        *u = *u + corr;
    }
    if (maxcorr) { // wave64 max of |corr|, one atomic per wave (non-negative floats order as uints)
        float m = fabsf(corr);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxcorr, __float_as_uint(m));
    }
}

void launch_prolong_add(Field Uc, Field Uf, MGGeom g, unsigned *d_maxcorr, hipStream_t s)
{
    dim3 grid((g.x.n + 63) / 64, (g.y.n + 3) / 4, Uf.C);
    hipLaunchKernelGGL(k_prolong_add, grid, dim3(256), 0, s, Uc, Uf, g, d_maxcorr);
}

} // namespace sc
