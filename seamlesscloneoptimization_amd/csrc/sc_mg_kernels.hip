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

// Thread = 4 consecutive fine points of one row (float4 read-modify-write of Uf).  With MAXC the
// largest |correction| of the launch is reduced wave64 shuffle -> LDS -> one plain store per
// block into `partial`, folded by k_max_final (a single atomic word would serialise ~10^5 waves).
template <bool MAXC>
__global__ __launch_bounds__(256) void k_prolong_add(Field Uc, Field Uf, MGGeom g, float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int y = blockIdx.y * 4 + wv + 1;
    const int c = blockIdx.z;
    float m = 0.f;
    if (y <= g.y.n && x <= g.x.n) {
        int J0, J1;
        float wy0, wy1;
        interp_1d(g.y, y, J0, J1, wy0, wy1);
        const float *__restrict__ e0 = Uc.at(c) + (size_t)J0 * Uc.pitch;
        const float *__restrict__ e1 = Uc.at(c) + (size_t)J1 * Uc.pitch;
        float *up = Uf.at(c) + (size_t)y * Uf.pitch + x;
        float4 u4 = *reinterpret_cast<float4 *>(up);
        float cr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xi = x + k;
            cr[k] = 0.f;
            if (xi >= 1 && xi <= g.x.n) {
                int I0, I1;
                float wx0, wx1;
                interp_1d(g.x, xi, I0, I1, wx0, wx1);
                const float top = wx0 * e0[I0] + wx1 * e0[I1];
                const float bot = wx0 * e1[I0] + wx1 * e1[I1];
                cr[k] = wy0 * top + wy1 * bot;
            }
        }
        if (x + 0 >= 1 && x + 0 <= g.x.n) u4.x = u4.x + cr[0];
        if (x + 1 <= g.x.n) u4.y = u4.y + cr[1];
        if (x + 2 <= g.x.n) u4.z = u4.z + cr[2];
        if (x + 3 <= g.x.n) u4.w = u4.w + cr[3];
        *reinterpret_cast<float4 *>(up) = u4;
        if (MAXC) m = fmaxf(fmaxf(fabsf(cr[0]), fabsf(cr[1])), fmaxf(fabsf(cr[2]), fabsf(cr[3])));
    }
    if (MAXC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float red[4];
        if (lane == 0) red[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
                fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

__global__ __launch_bounds__(256) void k_max_final(const float *__restrict__ partial, int n, unsigned *__restrict__ out)
{
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *out = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

int prolong_blocks(int nx, int ny, int C) { return ((nx + 1 + 255) / 256) * ((ny + 3) / 4) * C; }

void launch_prolong_add(Field Uc, Field Uf, MGGeom g, float *d_partial, unsigned *d_maxcorr, hipStream_t s)
{
    dim3 grid((g.x.n + 1 + 255) / 256, (g.y.n + 3) / 4, Uf.C);
    if (d_partial && d_maxcorr) {
        hipLaunchKernelGGL(k_prolong_add<true>, grid, dim3(256), 0, s, Uc, Uf, g, d_partial);
        hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, s, d_partial, (int)(grid.x * grid.y * grid.z), d_maxcorr);
    } else {
        hipLaunchKernelGGL(k_prolong_add<false>, grid, dim3(256), 0, s, Uc, Uf, g, (float *)nullptr);
    }
}

// ---- bottom of the V-cycle in ONE launch -----------------------------------------------------
// Every level small enough (<= MG_BOTTOM_POINTS unknowns per plane) is processed by a single
// 1024-thread workgroup per channel: smoothing, residual, restriction, coarsest solve,
// prolongation and post-smoothing run as block-strided loops separated by __syncthreads(), so
// the ~100 tiny launches those levels would need collapse into one.  Only this workgroup
// touches its channel's planes, so workgroup-scope visibility is all that is required.
__device__ __forceinline__ float gen_gs(const float *__restrict__ u, const float *__restrict__ f, int P, int x, int y,
                                        const MGGeom &g)
{
    const float cw = (x == g.x.n) ? g.x.cw_last : 1.0f, dx = (x == g.x.n) ? g.x.d_last : 2.0f;
    const float cn = (y == g.y.n) ? g.y.cw_last : 1.0f, dy = (y == g.y.n) ? g.y.d_last : 2.0f;
    const float *p = u + (size_t)y * P + x;
    return (((cw * p[-1] + p[1]) + (cn * p[-P] + p[P])) - f[(size_t)y * P + x]) / (dx + dy);
}

__device__ void bt_rb_half(float *u, const float *f, int P, const MGGeom &g, int color, float omega, bool sor)
{
    const int hx = (g.x.n + 1) / 2; // colour columns per row (upper bound)
    for (int i = threadIdx.x; i < hx * g.y.n; i += blockDim.x) {
        const int y = 1 + i / hx;
        const int x = 1 + 2 * (i - (y - 1) * hx) + ((1 + y + color) & 1);
        if (x > g.x.n) continue;
        const float gs = gen_gs(u, f, P, x, y, g);
        float *p = u + (size_t)y * P + x;
        *p = sor ? (*p + omega * (gs - *p)) : gs;
    }
    __syncthreads();
}

__device__ void bt_residual(const float *u, const float *f, float *r, int P, const MGGeom &g)
{
    for (int i = threadIdx.x; i < g.x.n * g.y.n; i += blockDim.x) {
        const int y = 1 + i / g.x.n, x = 1 + (i - (y - 1) * g.x.n);
        const double cw = (x == g.x.n) ? (double)g.x.cw_last : 1.0, dx = (x == g.x.n) ? (double)g.x.d_last : 2.0;
        const double cn = (y == g.y.n) ? (double)g.y.cw_last : 1.0, dy = (y == g.y.n) ? (double)g.y.d_last : 2.0;
        const float *p = u + (size_t)y * P + x;
        const double s = ((cw * (double)p[-1] + (double)p[1]) + (cn * (double)p[-P] + (double)p[P])) - (dx + dy) * (double)p[0];
        r[(size_t)y * P + x] = (float)((double)f[(size_t)y * P + x] - s);
    }
    __syncthreads();
}

__device__ void bt_restrict_zero(const float *r, int P, float *fc, float *uc, int Pc, const MGGeom &g)
{
    for (int i = threadIdx.x; i < g.x.nc * g.y.nc; i += blockDim.x) {
        const int J = 1 + i / g.x.nc, I = 1 + (i - (J - 1) * g.x.nc);
        float wx[4], wy[4], ix, iy;
        restrict_weights(g.x, I, wx, ix);
        restrict_weights(g.y, J, wy, iy);
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
        fc[(size_t)J * Pc + I] = 4.0f * (acc * (ix * iy));
        uc[(size_t)J * Pc + I] = 0.f;
    }
    __syncthreads();
}

__device__ void bt_prolong(const float *e, int Pc, float *u, int P, const MGGeom &g)
{
    for (int i = threadIdx.x; i < g.x.n * g.y.n; i += blockDim.x) {
        const int y = 1 + i / g.x.n, x = 1 + (i - (y - 1) * g.x.n);
        int I0, I1, J0, J1;
        float wx0, wx1, wy0, wy1;
        interp_1d(g.x, x, I0, I1, wx0, wx1);
        interp_1d(g.y, y, J0, J1, wy0, wy1);
        const float top = wx0 * e[(size_t)J0 * Pc + I0] + wx1 * e[(size_t)J0 * Pc + I1];
        const float bot = wx0 * e[(size_t)J1 * Pc + I0] + wx1 * e[(size_t)J1 * Pc + I1];
        float *p = u + (size_t)y * P + x;
        *p = *p + (wy0 * top + wy1 * bot);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_mg_bottom(MGBottomArgs a)
{
    const int c = blockIdx.x;
    const int L = a.nlevels;
    // the top bottom-level correction starts from zero (interior only; ring/pads are zero already)
    {
        const MGBottomLevel &t = a.lv[0];
        float *u = t.U.at(c);
        for (int i = threadIdx.x; i < t.g.x.n * t.g.y.n; i += blockDim.x) {
            const int y = 1 + i / t.g.x.n, x = 1 + (i - (y - 1) * t.g.x.n);
            u[(size_t)y * t.U.pitch + x] = 0.f;
        }
        __syncthreads();
    }
    for (int l = 0; l + 1 < L; ++l) {
        const MGBottomLevel &v = a.lv[l];
        const MGBottomLevel &w = a.lv[l + 1];
        float *u = v.U.at(c);
        const float *f = v.F.at(c);
        for (int s = 0; s < a.pre; ++s) {
            bt_rb_half(u, f, v.U.pitch, v.g, 0, 1.0f, false);
            bt_rb_half(u, f, v.U.pitch, v.g, 1, 1.0f, false);
        }
        bt_residual(u, f, v.T.at(c), v.U.pitch, v.g);
        bt_restrict_zero(v.T.at(c), v.U.pitch, w.F.at(c), w.U.at(c), w.U.pitch, v.g);
    }
    {
        const MGBottomLevel &v = a.lv[L - 1];
        for (int s = 0; s < a.coarse_sweeps; ++s) {
            bt_rb_half(v.U.at(c), v.F.at(c), v.U.pitch, v.g, 0, v.omega, true);
            bt_rb_half(v.U.at(c), v.F.at(c), v.U.pitch, v.g, 1, v.omega, true);
        }
    }
    for (int l = L - 2; l >= 0; --l) {
        const MGBottomLevel &v = a.lv[l];
        const MGBottomLevel &w = a.lv[l + 1];
        bt_prolong(w.U.at(c), w.U.pitch, v.U.at(c), v.U.pitch, v.g);
        for (int s = 0; s < a.post; ++s) {
            bt_rb_half(v.U.at(c), v.F.at(c), v.U.pitch, v.g, 0, 1.0f, false);
            bt_rb_half(v.U.at(c), v.F.at(c), v.U.pitch, v.g, 1, 1.0f, false);
        }
    }
}

void launch_mg_bottom(const MGBottomArgs &a, int C, hipStream_t s)
{
    hipLaunchKernelGGL(k_mg_bottom, dim3(C), dim3(1024), 0, s, a);
}

} // namespace sc
