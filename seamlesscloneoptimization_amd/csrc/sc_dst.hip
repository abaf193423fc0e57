// sc_dst.hip -- SC_METHOD_DST: the reference's own direct solve, hand-written for gfx950.
//
// The reference solves the 5-point system exactly like OpenCV: forward 2-D DST-I of the folded right-hand side, division
// by den[j][i] = filter_X[i] + filter_Y[j] - 4 with the FLOAT tables of seamlessClone_imp.cpp:596-599 added in float
// (:1651-1653), inverse DST-I, scaling (:1814-1896).  Its two back-ends form the transforms by batched 1-D FFTs of the odd
// extension (cuFFT, the default) or as four dense products with the DST matrix (cuBLAS, :1266-1334):
//     u = S_h ( (S_h g S_w) / den ) S_w * 4 / ((w+1)(h+1)),        S_n[i][j] = sin(pi (i+1)(j+1) / (n+1)).
// This file is the matrix form on the MI355X matrix cores -- v_mfma_f64_16x16x4_f64, LDS-tiled 128 x 128 x 16, one
// workgroup per tile and channel -- with everything between the float input and the float output in double, so the only
// float32 effects left are the ones that define the reference's answer: the float tables and the float sum in `den`.
// (The reference's own matrix back-end builds S with the float literal PI, which costs it up to 6 grey levels at
// 2400 x 1552, PDF p3; its FFT back-end -- the default, and OpenCV -- has exact transforms up to float32 rounding.  The
// oracle's transforms are double inside as well: oracle/sc_oracle.c.)
// O(n^3): 0.2 TFLOP for a 2048^2 ROI, 1.6 TFLOP at 4096^2 -- milliseconds where the multigrid path takes a fraction of
// one; it exists as the non-iterative cross-check of the default path (which reaches the same answer through
// sc_lowmode.hip) and for callers who want the reference's arithmetic with nothing iterative in it.
#include "sc_instance.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace sc {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int DG_BM = 128, DG_BN = 128, DG_BK = 16, DG_PAD = 2;

// The DST matrix S[i][j] = sin(pi (i+1)(j+1)/(n+1)) is symmetric, and S[i][n-1-j] = (-1)^i S[i][j]: modes with even index are
// symmetric about the middle of the interval, modes with odd index antisymmetric.  Folding a vector into its symmetric and
// antisymmetric halves (m = ceil(n/2) entries each) therefore splits a product with S into two products of half the size --
// half the multiply-adds of the plain matrix form.  Four m x m blocks per direction (padded to mp, row-major, zero padded):
//   T[0] = Se [x][j'] = S[x][2j']      T[1] = So [x][j'] = S[x][2j'+1]      (x < m: the folded index)
//   T[2] = Se^T[j'][x] = S[2j'][x]     T[3] = So^T[j'][x] = S[2j'+1][x]
__global__ __launch_bounds__(256) void k_dst_table(double *__restrict__ T, int n, int mp)
{
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long blk = (long)mp * mp;
    if (id >= 4 * blk) return;
    const int which = (int)(id / blk), r = (int)((id % blk) / mp), c = (int)(id % mp);
    const int m = (n + 1) / 2;
    const int x = (which < 2) ? r : c, jp = (which < 2) ? c : r;       // folded index, mode-pair index
    const int j = 2 * jp + (which & 1);
    double v = 0.0;
    if (x < m && j < n) {
        const long per = 2L * (n + 1);
        const long q = ((long)(x + 1) * (j + 1)) % per;
        v = sinpi((double)q / (double)(n + 1));
    }
    T[id] = v;
}

// Folded right-hand side, double: g = lap - ring neighbours (seamlessClone_imp.cpp:1992-2008), then
// G[py mph + y][px mpw + x] = sum over the (up to) four mirror images of (y, x) with the signs of the (py, px) parity class;
// a middle row / column (odd size) is its own mirror image: counted once in the symmetric half, zero in the antisymmetric.
__device__ __forceinline__ float dst_g(const Field &U, const Field &F, int c, int x, int y)
{
    const int w = U.W - 2, h = U.H - 2;
    const size_t o = (size_t)(y + 1) * U.pitch + (x + 1);
    const float *__restrict__ u = U.at(c);
    float g = F.at(c)[o];
    if (x == 0) g -= u[o - 1];
    if (y == 0) g -= u[o - U.pitch];
    if (x == w - 1) g -= u[o + 1];
    if (y == h - 1) g -= u[o + U.pitch];
    return g;
}

__global__ __launch_bounds__(256) void k_dst_fold(Field U, Field F, double *__restrict__ G, int mph, int mpw)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;      // folded coordinates
    if (x >= mpw) return;
    const int w = U.W - 2, h = U.H - 2, mw = (w + 1) / 2, mh = (h + 1) / 2;
    double v00 = 0.0, v01 = 0.0, v10 = 0.0, v11 = 0.0;
    if (x < mw && y < mh) {
        const int xm = w - 1 - x, ym = h - 1 - y;
        const bool xs = xm != x, ys = ym != y;                       // a true mirror image exists
        const double a = dst_g(U, F, c, x, y), b = xs ? dst_g(U, F, c, xm, y) : 0.0;
        const double d = ys ? dst_g(U, F, c, x, ym) : 0.0, e = (xs && ys) ? dst_g(U, F, c, xm, ym) : 0.0;
        const double se = a + b, so = xs ? a - b : 0.0;              // row y: symmetric / antisymmetric in x
        const double te = d + e, to = xs ? d - e : 0.0;              // mirror row
        v00 = se + te; v01 = so + to;
        v10 = ys ? se - te : 0.0; v11 = ys ? so - to : 0.0;
    }
    double *__restrict__ g = G + (size_t)c * (2 * mph) * (2 * mpw);
    g[(size_t)y * (2 * mpw) + x] = v00;               g[(size_t)y * (2 * mpw) + mpw + x] = v01;
    g[(size_t)(mph + y) * (2 * mpw) + x] = v10;       g[(size_t)(mph + y) * (2 * mpw) + mpw + x] = v11;
}

// The inverse of the fold on the result: the four parity blocks R[p][q] of the last product give the field at (y, x) and
// its three mirror images; written as float, scaled, into the interior of the planar field.
__global__ __launch_bounds__(256) void k_dst_unfold(const double *__restrict__ R, int mph, int mpw, double scale, Field U)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    const int w = U.W - 2, h = U.H - 2, mw = (w + 1) / 2, mh = (h + 1) / 2;
    if (x >= mw || y >= mh) return;
    const double *__restrict__ r = R + (size_t)c * (2 * mph) * (2 * mpw);
    const double b00 = r[(size_t)y * (2 * mpw) + x], b01 = r[(size_t)y * (2 * mpw) + mpw + x];
    const double b10 = r[(size_t)(mph + y) * (2 * mpw) + x], b11 = r[(size_t)(mph + y) * (2 * mpw) + mpw + x];
    float *__restrict__ u = U.at(c);
    const int xm = w - 1 - x, ym = h - 1 - y;
    u[(size_t)(y + 1) * U.pitch + x + 1] = (float)(((b00 + b01) + (b10 + b11)) * scale);
    if (xm != x) u[(size_t)(y + 1) * U.pitch + xm + 1] = (float)(((b00 - b01) + (b10 - b11)) * scale);
    if (ym != y) u[(size_t)(ym + 1) * U.pitch + x + 1] = (float)(((b00 + b01) - (b10 + b11)) * scale);
    if (xm != x && ym != y) u[(size_t)(ym + 1) * U.pitch + xm + 1] = (float)(((b00 - b01) - (b10 - b11)) * scale);
}

// C = A * B in double on the matrix cores.  Row-major, every dimension padded (M, N to 128, K to 16): no edge handling in
// the main loop.  blockIdx.z = 2 * channel + parity half: each operand has a stride per channel and one per half (0 shares
// the table blocks between channels).
// EPI 0: C double.  EPI 1: C double, divided by the reference's float denominator fx[j] + fy[i] - 4, where (i, j) are the
// MODE indices of the element -- rows / columns are in parity-split order: index r of a direction with half size mp is
// mode 2 (r % mp) + r / mp (zero outside eh x ew).

template <int EPI>
__global__ __launch_bounds__(512) void k_dgemm(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                               int lda, int ldb, int ldc, int K, size_t strideAc, size_t strideAp, size_t strideBc,
                                               size_t strideBp, size_t strideCc, size_t strideCp,
                                               const float *__restrict__ fx, const float *__restrict__ fy, int eh, int ew, int mph, int mpw,
                                               int exact_den)
{
    // Eight waves, two per SIMD (their MFMAs interleave in the pipe), each owns a 32 x 64 part of the 128 x 128 tile: 2 x 4 MFMA tiles.
    // Two LDS stages: while stage s is multiplied, the next K-tile (already in registers) is written to stage s^1 -- one barrier per K-tile.
    __shared__ double As[2][DG_BK][DG_BM + DG_PAD];
    __shared__ double Bs[2][DG_BK][DG_BN + DG_PAD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;       // wm 0..3 (32 rows each), wn 0..1 (64 columns each)
    const int m0 = blockIdx.y * DG_BM, n0 = blockIdx.x * DG_BN, zc = blockIdx.z >> 1, zp = blockIdx.z & 1;
    A += (size_t)zc * strideAc + (size_t)zp * strideAp; B += (size_t)zc * strideBc + (size_t)zp * strideBp;
    C += (size_t)zc * strideCc + (size_t)zp * strideCp;
    // global -> register staging: A tile 128 x 16 (thread: one row, 4 consecutive k), B tile 16 x 128 (one k row, 4 columns)
    const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 5, bc = (t & 31) * 4;
    const double *__restrict__ ap = A + (size_t)(m0 + ar) * lda + ak;
    const double *__restrict__ bp = B + (size_t)bk * ldb + n0 + bc;
    double2 ra0, ra1, rb0, rb1;
#define DG_LOAD(k0)                                                                                          \
    {                                                                                                        \
        const double *a_ = ap + (k0);                                                                        \
        const double *b_ = bp + (size_t)(k0) * ldb;                                                          \
        ra0 = *reinterpret_cast<const double2 *>(a_); ra1 = *reinterpret_cast<const double2 *>(a_ + 2);      \
        rb0 = *reinterpret_cast<const double2 *>(b_); rb1 = *reinterpret_cast<const double2 *>(b_ + 2);      \
    }
#define DG_STORE(s)                                                                                          \
    {                                                                                                        \
        As[s][ak + 0][ar] = ra0.x; As[s][ak + 1][ar] = ra0.y; As[s][ak + 2][ar] = ra1.x; As[s][ak + 3][ar] = ra1.y;  \
        *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 0]) = rb0; *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 2]) = rb1;  \
    }
    v4f64 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){ 0.0, 0.0, 0.0, 0.0 };
    const int fr = lane & 15, fk = lane >> 4;
    DG_LOAD(0);
    DG_STORE(0);
    __syncthreads();
    const int ntiles = K / DG_BK;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int s = kt & 1;
        // the next tile's loads fly while this one is multiplied (the last iteration re-reads the last tile: no branch)
        DG_LOAD(min(kt + 1, ntiles - 1) * DG_BK);
#pragma unroll
        for (int kk = 0; kk < DG_BK; kk += 4) {
            double a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[s][kk + fk][wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[s][kk + fk][wn * 64 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the stores BEHIND the products: the compiler otherwise hoists them (and the wait for the loads) above
        DG_STORE(s ^ 1);                 // stage s^1 was last read in iteration kt-1: every wave passed the barrier below since
        __syncthreads();
    }
#undef DG_LOAD
#undef DG_STORE
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + fk + 4 * r, col = n0 + wn * 64 + j * 16 + fr;
                double v = acc[i][j][r];
                if (EPI == 1) {
                    // this launch is a LEFT product: its rows are one parity half (zp) of the modes in y, its columns run over both halves in x
                    const int i = 2 * row + zp, j = 2 * (col % mpw) + col / mpw;
                    const bool in = row < mph && i < eh && j < ew;
                    if (exact_den) {           // the float tables are singular at this size (the reference divides by zero): exact system
                        const double sa = sinpi(0.5 * (j + 1.0) / (ew + 1.0)), sb = sinpi(0.5 * (i + 1.0) / (eh + 1.0));
                        v = in ? v / (-4.0 * (sa * sa + sb * sb)) : 0.0;
                    } else {
                        const float den = in ? (fx[j] + fy[i]) - 4.0f : 1.0f;       // seamlessClone_imp.cpp:1651-1653, in float
                        v = in ? v / (double)den : 0.0;
                    }
                }
                C[(size_t)row * ldc + col] = v;
            }
}

// ---------------------------------------------------------------------------------------------------------- host side

static int dst_prepare(Instance *I)
{
    DstState &D = I->dst;
    const int w = I->F.W - 2, h = I->F.H - 2, C = I->F.C;
    const int mpw = round_up((w + 1) / 2, 128), mph = round_up((h + 1) / 2, 128);      // padded half sizes
    int rc;
    const size_t plane = (size_t)(2 * mph) * (2 * mpw) * sizeof(double);
    for (DevBuf *b : { &D.G, &D.T1, &D.T2 })
        if ((rc = ensure(I, *b, plane * C))) return rc;
    if (D.w == w && D.h == h && D.Sw.p && D.Sh.p) return SC_OK;
    I->info.new_size = 1;
    if ((rc = ensure(I, D.Sw, (size_t)4 * mpw * mpw * sizeof(double)))) return rc;
    if ((rc = ensure(I, D.Sh, (size_t)4 * mph * mph * sizeof(double)))) return rc;
    if ((rc = ensure(I, D.fxy, (size_t)(w + h) * sizeof(float)))) return rc;
    if ((rc = ensure_pinned(I, D.hfxy, (size_t)(w + h) * sizeof(float)))) return rc;
    // the reference's float tables (seamlessClone_imp.cpp:596-599; PI is the float literal of seamlessClone_imp.h:17)
    const double PIf = (double)3.14159265358979323846f;
    float *fx = (float *)D.hfxy.p, *fy = fx + w;
    for (int i = 0; i < w; ++i) fx[i] = (float)(2.0 * std::cos(PIf / (w + 1.0) * (i + 1.0)));
    for (int j = 0; j < h; ++j) fy[j] = (float)(2.0 * std::cos(PIf / (h + 1.0) * (j + 1.0)));
    D.singular = !((fx[0] + fy[0]) - 4.0f < 0.0f);     // 2 cos(pi/(n+1)) rounds to 2.0f in both directions (n >= ~12 870): the reference divides by zero
    SC_HIP(I, hipMemcpyAsync(D.fxy.p, D.hfxy.p, (size_t)(w + h) * sizeof(float), hipMemcpyHostToDevice, I->stream));
    hipLaunchKernelGGL(k_dst_table, dim3((unsigned)(((size_t)4 * mpw * mpw + 255) / 256)), dim3(256), 0, I->stream, (double *)D.Sw.p, w, mpw);
    hipLaunchKernelGGL(k_dst_table, dim3((unsigned)(((size_t)4 * mph * mph + 255) / 256)), dim3(256), 0, I->stream, (double *)D.Sh.p, h, mph);
    SC_HIP(I, hipGetLastError());
    D.w = w; D.h = h; D.wp = mpw; D.hp = mph;
    return SC_OK;
}

// Direct solve of the fields bound to the instance: interior of result(I) <- the reference's answer.  F must be float.
int dst_solve(Instance *I)
{
    if (I->f_half) { I->err = "internal: float16 right-hand side on the direct path"; return SC_ERR_BAD_ARG; }
    int rc = dst_prepare(I);
    if (rc) return rc;
    DstState &D = I->dst;
    const int C = I->F.C, mpw = D.wp, mph = D.hp;
    Field &U = I->result_in_U1 ? I->U1 : I->U0;
    double *G = (double *)D.G.p, *T1 = (double *)D.T1.p, *T2 = (double *)D.T2.p;
    const double *Tw = (const double *)D.Sw.p, *Th = (const double *)D.Sh.p;
    const size_t plane = (size_t)(2 * mph) * (2 * mpw), bw = (size_t)mpw * mpw, bh = (size_t)mph * mph;
    const int ld = 2 * mpw;
    const float *fx = (const float *)D.fxy.p, *fy = fx + D.w;
    const double scale = 4.0 / ((D.w + 1.0) * (D.h + 1.0));
    hipLaunchKernelGGL(k_dst_fold, dim3((mpw + 255) / 256, mph, C), dim3(256), 0, I->stream, U, I->F, G, mph, mpw);
    // Right products work on one column half (all 2 mph rows), left products on one row half (all 2 mpw columns); the half is
    // the low bit of blockIdx.z.   forward: T1 = G [Se|So]_w ; T2 = ([Se^T;So^T]_h T1) / den     inverse: T1 = T2 [Se^T|So^T]_w ; T2 = [Se;So]_h T1
    const dim3 gr(mpw / DG_BN, 2 * mph / DG_BM, 2 * C), gl(2 * mpw / DG_BN, mph / DG_BM, 2 * C);
#define DG_EPI fx, fy, D.h, D.w, mph, mpw, (D.singular || (I->opts.flags & SC_FLAG_EXACT_TABLES)) ? 1 : 0
    hipLaunchKernelGGL(k_dgemm<0>, gr, dim3(512), 0, I->stream, G, Tw, T1, ld, mpw, ld, mpw, plane, (size_t)mpw, (size_t)0, bw, plane, (size_t)mpw, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<1>, gl, dim3(512), 0, I->stream, Th + 2 * bh, T1, T2, mph, ld, ld, mph, (size_t)0, bh, plane, (size_t)mph * ld, plane, (size_t)mph * ld, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<0>, gr, dim3(512), 0, I->stream, T2, Tw + 2 * bw, T1, ld, mpw, ld, mpw, plane, (size_t)mpw, (size_t)0, bw, plane, (size_t)mpw, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<0>, gl, dim3(512), 0, I->stream, Th, T1, T2, mph, ld, ld, mph, (size_t)0, bh, plane, (size_t)mph * ld, plane, (size_t)mph * ld, DG_EPI);
#undef DG_EPI
    hipLaunchKernelGGL(k_dst_unfold, dim3(((D.w + 1) / 2 + 255) / 256, (D.h + 1) / 2, C), dim3(256), 0, I->stream, (const double *)T2, mph, mpw, scale, U);
    SC_HIP(I, hipGetLastError());
    I->info.sweeps = 1;
    I->info.converged = 1;
    I->info.sweep_launches += 4;
    return SC_OK;
}

} // namespace sc
