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

// S[i][j] = sin(pi (i+1)(j+1) / (n+1)) for i, j < n, zero in the padding (np x np, row-major)
__global__ __launch_bounds__(256) void k_dst_table(double *__restrict__ S, int n, int np)
{
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)np * np) return;
    const int i = (int)(id / np), j = (int)(id % np);
    double v = 0.0;
    if (i < n && j < n) {
        const long m = 2L * (n + 1);
        const long q = ((long)(i + 1) * (j + 1)) % m;
        v = sinpi((double)q / (double)(n + 1));
    }
    S[id] = v;
}

// folded right-hand side as double, zero padded: g = lap - ring neighbours (seamlessClone_imp.cpp:1992-2008)
__global__ __launch_bounds__(256) void k_dst_fold(Field U, Field F, double *__restrict__ G, int hp, int wp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;      // padded interior coordinates
    if (x >= wp) return;
    const int w = U.W - 2, h = U.H - 2;
    double v = 0.0;
    if (x < w && y < h) {
        const size_t o = (size_t)(y + 1) * U.pitch + (x + 1);
        const float *__restrict__ u = U.at(c);
        float g = F.at(c)[o];
        if (x == 0) g -= u[o - 1];
        if (y == 0) g -= u[o - U.pitch];
        if (x == w - 1) g -= u[o + 1];
        if (y == h - 1) g -= u[o + U.pitch];
        v = (double)g;
    }
    G[((size_t)c * hp + y) * wp + x] = v;
}

// C = A * B in double on the matrix cores.  Row-major, every dimension padded (M, N to 128, K to 16): no edge handling in
// the main loop.  blockIdx.z = channel; a stride of 0 shares an operand (the DST matrix) between channels.
// EPI 0: C double.  EPI 1: C double, divided by the reference's float denominator fx[col] + fy[row] - 4 (zero outside
// eh x ew).  EPI 2: C * scale as float into the interior of the planar field Uf (row r -> field row r+1).

template <int EPI>
__global__ __launch_bounds__(256) void k_dgemm(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                               int lda, int ldb, int ldc, int K, size_t strideA, size_t strideB, size_t strideC,
                                               const float *__restrict__ fx, const float *__restrict__ fy, int eh, int ew, double scale,
                                               float *__restrict__ Uf, int upitch, size_t uplane)
{
    // two LDS stages: while stage s is multiplied, the next K-tile (already in registers) is written to stage s^1 -- one
    // barrier per K-tile
    __shared__ double As[2][DG_BK][DG_BM + DG_PAD];
    __shared__ double Bs[2][DG_BK][DG_BN + DG_PAD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * DG_BM, n0 = blockIdx.x * DG_BN, z = blockIdx.z;
    A += (size_t)z * strideA; B += (size_t)z * strideB;
    // global -> register staging: A tile 128 x 16 (thread: one row, 8 consecutive k), B tile 16 x 128 (one k row, 8 columns)
    const int ar = t >> 1, ak = (t & 1) * 8, bk = t >> 4, bc = (t & 15) * 8;
    const double *__restrict__ ap = A + (size_t)(m0 + ar) * lda + ak;
    const double *__restrict__ bp = B + (size_t)bk * ldb + n0 + bc;
    double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define DG_LOAD(k0)                                                                                          \
    {                                                                                                        \
        const double *a_ = ap + (k0);                                                                        \
        const double *b_ = bp + (size_t)(k0) * ldb;                                                          \
        ra0 = *reinterpret_cast<const double2 *>(a_); ra1 = *reinterpret_cast<const double2 *>(a_ + 2);      \
        ra2 = *reinterpret_cast<const double2 *>(a_ + 4); ra3 = *reinterpret_cast<const double2 *>(a_ + 6);  \
        rb0 = *reinterpret_cast<const double2 *>(b_); rb1 = *reinterpret_cast<const double2 *>(b_ + 2);      \
        rb2 = *reinterpret_cast<const double2 *>(b_ + 4); rb3 = *reinterpret_cast<const double2 *>(b_ + 6);  \
    }
#define DG_STORE(s)                                                                                          \
    {                                                                                                        \
        As[s][ak + 0][ar] = ra0.x; As[s][ak + 1][ar] = ra0.y; As[s][ak + 2][ar] = ra1.x; As[s][ak + 3][ar] = ra1.y;  \
        As[s][ak + 4][ar] = ra2.x; As[s][ak + 5][ar] = ra2.y; As[s][ak + 6][ar] = ra3.x; As[s][ak + 7][ar] = ra3.y;  \
        *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 0]) = rb0; *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 2]) = rb1;  \
        *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 4]) = rb2; *reinterpret_cast<double2 *>(&Bs[s][bk][bc + 6]) = rb3;  \
    }
    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
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
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[s][kk + fk][wm * 64 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[s][kk + fk][wn * 64 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; ++i)
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + fk + 4 * r, col = n0 + wn * 64 + j * 16 + fr;
                double v = acc[i][j][r];
                if (EPI == 1) {
                    const bool in = row < eh && col < ew;
                    const float den = in ? (fx[col] + fy[row]) - 4.0f : 1.0f;       // seamlessClone_imp.cpp:1651-1653, in float
                    v = in ? v / (double)den : 0.0;
                }
                if (EPI == 2) {
                    if (row < eh && col < ew) Uf[(size_t)z * uplane + (size_t)(row + 1) * upitch + col + 1] = (float)(v * scale);
                } else {
                    C[(size_t)z * strideC + (size_t)row * ldc + col] = v;
                }
            }
}

// ---------------------------------------------------------------------------------------------------------- host side

static int dst_prepare(Instance *I)
{
    DstState &D = I->dst;
    const int w = I->F.W - 2, h = I->F.H - 2, C = I->F.C;
    const int wp = round_up(w, 128), hp = round_up(h, 128);
    int rc;
    const size_t plane = (size_t)hp * wp * sizeof(double);
    for (DevBuf *b : { &D.G, &D.T1, &D.T2 })
        if ((rc = ensure(I, *b, plane * C))) return rc;
    if (D.w == w && D.h == h && D.Sw.p && D.Sh.p) return SC_OK;
    if ((rc = ensure(I, D.Sw, (size_t)wp * wp * sizeof(double)))) return rc;
    if ((rc = ensure(I, D.Sh, (size_t)hp * hp * sizeof(double)))) return rc;
    if ((rc = ensure(I, D.fxy, (size_t)(wp + hp) * sizeof(float)))) return rc;
    if ((rc = ensure_pinned(I, D.hfxy, (size_t)(wp + hp) * sizeof(float)))) return rc;
    // the reference's float tables (seamlessClone_imp.cpp:596-599; PI is the float literal of seamlessClone_imp.h:17)
    const double PIf = (double)3.14159265358979323846f;
    float *fx = (float *)D.hfxy.p, *fy = fx + wp;
    for (int i = 0; i < wp; ++i) fx[i] = i < w ? (float)(2.0 * std::cos(PIf / (w + 1.0) * (i + 1.0))) : 0.f;
    for (int j = 0; j < hp; ++j) fy[j] = j < h ? (float)(2.0 * std::cos(PIf / (h + 1.0) * (j + 1.0))) : 0.f;
    SC_HIP(I, hipMemcpyAsync(D.fxy.p, D.hfxy.p, (size_t)(wp + hp) * sizeof(float), hipMemcpyHostToDevice, I->stream));
    hipLaunchKernelGGL(k_dst_table, dim3((unsigned)(((size_t)wp * wp + 255) / 256)), dim3(256), 0, I->stream, (double *)D.Sw.p, w, wp);
    hipLaunchKernelGGL(k_dst_table, dim3((unsigned)(((size_t)hp * hp + 255) / 256)), dim3(256), 0, I->stream, (double *)D.Sh.p, h, hp);
    SC_HIP(I, hipGetLastError());
    D.w = w; D.h = h; D.wp = wp; D.hp = hp;
    return SC_OK;
}

// Direct solve of the fields bound to the instance: interior of result(I) <- the reference's answer.  F must be float.
int dst_solve(Instance *I)
{
    if (I->f_half) { I->err = "internal: float16 right-hand side on the direct path"; return SC_ERR_BAD_ARG; }
    int rc = dst_prepare(I);
    if (rc) return rc;
    DstState &D = I->dst;
    const int C = I->F.C, wp = D.wp, hp = D.hp;
    Field &U = I->result_in_U1 ? I->U1 : I->U0;
    double *G = (double *)D.G.p, *T1 = (double *)D.T1.p, *T2 = (double *)D.T2.p;
    const double *Sw = (const double *)D.Sw.p, *Sh = (const double *)D.Sh.p;
    const size_t plane = (size_t)hp * wp;
    const float *fx = (const float *)D.fxy.p, *fy = fx + wp;
    const double scale = 4.0 / ((D.w + 1.0) * (D.h + 1.0));
    const dim3 grid(wp / DG_BN, hp / DG_BM, C);
    hipLaunchKernelGGL(k_dst_fold, dim3((wp + 255) / 256, hp, C), dim3(256), 0, I->stream, U, I->F, G, hp, wp);
    // T1 = G Sw ; T2 = (Sh T1) / den ; T1 = T2 Sw ; U = Sh T1 * scale
#define DG_EPI fx, fy, D.h, D.w, scale, U.p, U.pitch, U.plane
    hipLaunchKernelGGL(k_dgemm<0>, grid, dim3(256), 0, I->stream, G, Sw, T1, wp, wp, wp, wp, plane, (size_t)0, plane, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<1>, grid, dim3(256), 0, I->stream, Sh, T1, T2, hp, wp, wp, hp, (size_t)0, plane, plane, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<0>, grid, dim3(256), 0, I->stream, T2, Sw, T1, wp, wp, wp, wp, plane, (size_t)0, plane, DG_EPI);
    hipLaunchKernelGGL(k_dgemm<2>, grid, dim3(256), 0, I->stream, Sh, T1, T2, hp, wp, wp, hp, (size_t)0, plane, plane, DG_EPI);
#undef DG_EPI
    SC_HIP(I, hipGetLastError());
    I->info.sweeps = 1;
    I->info.converged = 1;
    I->info.sweep_launches += 4;
    return SC_OK;
}

} // namespace sc
