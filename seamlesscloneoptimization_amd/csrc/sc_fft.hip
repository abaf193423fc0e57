// sc_fft.hip -- SC_METHOD_FFT: the reference's DEFAULT direct back-end (poissonSolver2D_FFT, seamlessClone_imp.cpp:1694-1918),
// hand-written for gfx950: O(n^2 log n), float32 like cuFFT and OpenCV's dft.
//
// The reference solves the 5-point system with two batched 1-D complex FFTs of the odd extension per direction (length
// 2n+2, :1337-1456 / :1735-1794), divides by den = filter_X + filter_Y - 4 (float tables, :1642-1669) and transforms back
// (:1814-1896).  The transform it needs is the DST-I,
//     X_k = sum_{j=1..n} x_j sin(pi j k / (n+1)),   k = 1..n,
// and 2n+2 is rarely a friendly FFT length (4094 = 2 x 23 x 89 at the headline size; cuFFT itself goes through Bluestein for
// such lengths).  This file computes the DST-I directly as a chirp-z transform: with N = 2(n+1), c_m = exp(i pi m^2 / N),
//     sin(2 pi j k / N) = Im[ c_j c_k conj(c_{k-j}) ]      =>      X_k = Im[ c_k  sum_j (x_j c_j) conj(c_{k-j}) ],
// a linear convolution of n points with a 2n-1 point kernel, i.e. a circular convolution of any length M >= 2n-1: M is the
// next power of two (4096 for n = 2046 -- HALF the 8192 the odd-extension route would need), and the convolution is two
// power-of-two FFTs in LDS around a pointwise product with the precomputed transform of the chirp.
//
// (Measured and not kept: pass 0 of the forward / adjoint transform working straight between global memory and registers, and
// the last forward pass fused with the pointwise product and the first adjoint pass -- half the LDS round trips and barriers,
// but 146-315 VGPRs instead of 87-136 and only M/16 threads busy in the end passes: 5-20 % faster from 2048^2 up, 15-40 % SLOWER
// below 1024^2, which is where this path is the default.)
// One workgroup per row: the row is loaded once, multiplied by the chirp into LDS (M complex floats, padded against bank
// conflicts), transformed by in-place radix-16 / 8 / 4 / 2 Gentleman-Sande passes (registers hold a 16-point DFT; the
// result stays in digit-reversed order), multiplied by the chirp's transform (stored in that same order, 1/M folded in),
// and brought back by the adjoint passes in reverse order -- no reordering pass at all.  Twiddles come from a table computed
// in double; the chirp phases are reduced exactly in integers (m^2 mod 2N) before the double sincospi.
//
// The 2-D solve is three such launches and two tiled transposes:
//     rows (fold of the Dirichlet ring fused into the load) -> transpose -> columns: DST, divide by den, DST again, all in
//     one launch (the row never leaves LDS) -> transpose -> rows (scale 4/((w+1)(h+1)) and the store into the field fused).
// Everything is float32 (the reference's precision); the float tables are the reference's to the letter (:596-599).
#include "sc_instance.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace sc {

constexpr int FFT_THREADS = 256;
constexpr int FFT_MAX_LOGM = 14;                       // M = 16384: n <= 8192 unknowns per side (139 KB of LDS)
__host__ __device__ __forceinline__ int fft_pad(int i) { return i + (i >> 4); }     // 1 pad element per 16: keeps the strided passes off one bank

// T = float: the reference's precision (cuFFT, OpenCV's dft).  T = double (SC_FLAG_FFT_FP64): the same transforms in double --
// twice the LDS traffic, worth it for small ROIs where the launches are latency bound anyway: the result then carries no
// transform rounding at all (diff sum against the float-table port 2 instead of ~50 at the reference's 300x194 patch).
template <typename T> struct cx2 { T x, y; };
template <typename T> __host__ __device__ __forceinline__ cx2<T> mk(T x, T y) { cx2<T> r; r.x = x; r.y = y; return r; }

template <typename T>
struct FftPlan {                                       // one direction (by value in the kernel arguments)
    const cx2<T> *chirp;                               // c_m, m = 0 .. n
    const cx2<T> *bhat;                                // transform of the chirp kernel, digit-reversed order, scaled by 1/M
    const cx2<T> *tw;                                  // exp(-2 pi i k / M), k = 0 .. M-1
    const cx2<T> *tw2;                                 // exp(-2 pi i k / 2^logM), k = 0 .. 2^logM - 1 (== tw when M is a power of two)
    int n, M, logM, npass;                             // M = r 2^logM, r = 1, 3 or 5 (round 5: mixed-radix lengths)
    int r;
    int lr[4];                                         // log2 of the radix of each forward power-of-two pass
};

template <typename T> __device__ __forceinline__ cx2<T> cmulf(cx2<T> a, cx2<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __device__ __forceinline__ cx2<T> cmulcf(cx2<T> a, cx2<T> b) { return mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// R-point DFT in registers, natural order in and out; SGN = -1: exp(-2 pi i jk/R), +1: the conjugate.
// Radix-2 decimation-in-frequency stages with compile-time twiddles, then the bit-reversal as register renaming.
template <int R, int SGN, typename T>
__device__ __forceinline__ void dft_small(cx2<T> (&v)[R])
{
    constexpr double C16[8] = { 1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0, -0.38268343236508977,
                                -0.70710678118654752, -0.92387953251128674 };
    constexpr double S16[8] = { 0.0, 0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1.0, 0.92387953251128674,
                                0.70710678118654752, 0.38268343236508977 };
#pragma unroll
    for (int len = R; len >= 2; len >>= 1) {
        const int half = len >> 1;
#pragma unroll
        for (int blk = 0; blk < R; blk += len) {
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const cx2<T> a = v[blk + k], b = v[blk + k + half];
                v[blk + k] = mk<T>(a.x + b.x, a.y + b.y);
                const cx2<T> d = mk<T>(a.x - b.x, a.y - b.y);
                const int e = k * (16 / len);                   // twiddle exp(SGN 2 pi i e / 16), e in 0..7
                if (e == 0) v[blk + k + half] = d;
                else if (e == 4) v[blk + k + half] = (SGN < 0) ? mk<T>(d.y, -d.x) : mk<T>(-d.y, d.x);
                else {
                    const T c = (T)C16[e], s = (T)((SGN < 0) ? -S16[e] : S16[e]);
                    v[blk + k + half] = mk<T>(d.x * c - d.y * s, d.x * s + d.y * c);
                }
            }
        }
    }
    if (R > 2) {
        cx2<T> t[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int r = 0;
#pragma unroll
            for (int bit = 1, rb = R >> 1; bit < R; bit <<= 1, rb >>= 1) if (i & bit) r |= rb;
            t[r] = v[i];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = t[i];
    }
}

// Twiddles w^q, q = 1 .. R-1, of one butterfly (w = w_L^t): the powers of two come from the table (1, 2, 4, 8: at most four
// gathers), the others are products of two of them (at most three multiplications deep).  Fifteen gathers per 16-point
// butterfly made the passes wait for the L1, not for the LDS.
template <int R, typename T>
__device__ __forceinline__ void twiddle_powers(cx2<T> (&w)[R], const cx2<T> *__restrict__ tw, int t, int twsh, int M)
{
#pragma unroll
    for (int q = 1; q < R; ++q) {
        if ((q & (q - 1)) == 0) w[q] = tw[((t * q) << twsh) & (M - 1)];
        else {
            int hb = 1;
#pragma unroll
            for (int b = 1; b < R; b <<= 1) if (b <= q) hb = b;
            w[q] = cmulf<T>(w[hb], w[q - hb]);
        }
    }
}

// 3- and 5-point DFTs in registers, natural order in and out (the odd factor of a mixed-radix length M = 3 2^k or 5 2^k)
template <int RR, int SGN, typename T>
__device__ __forceinline__ void dft_odd(cx2<T> (&v)[RR])
{
    if constexpr (RR == 3) {
        const T S3 = (T)0.86602540378443865;
        const cx2<T> t1 = mk<T>(v[1].x + v[2].x, v[1].y + v[2].y);
        const cx2<T> t2 = mk<T>(v[0].x - (T)0.5 * t1.x, v[0].y - (T)0.5 * t1.y);
        const cx2<T> t3 = mk<T>(S3 * (v[1].x - v[2].x), S3 * (v[1].y - v[2].y));
        v[0] = mk<T>(v[0].x + t1.x, v[0].y + t1.y);
        if (SGN < 0) { v[1] = mk<T>(t2.x + t3.y, t2.y - t3.x); v[2] = mk<T>(t2.x - t3.y, t2.y + t3.x); }      // t2 -+ i t3
        else         { v[1] = mk<T>(t2.x - t3.y, t2.y + t3.x); v[2] = mk<T>(t2.x + t3.y, t2.y - t3.x); }
    } else {
        static_assert(RR == 5, "odd factors: 3 and 5");
        const T C1 = (T)0.30901699437494742, C2 = (T)-0.80901699437494742, S1 = (T)0.95105651629515357, S2 = (T)0.58778525229247313;
        const cx2<T> t1 = mk<T>(v[1].x + v[4].x, v[1].y + v[4].y), t2 = mk<T>(v[2].x + v[3].x, v[2].y + v[3].y);
        const cx2<T> t3 = mk<T>(v[1].x - v[4].x, v[1].y - v[4].y), t4 = mk<T>(v[2].x - v[3].x, v[2].y - v[3].y);
        const cx2<T> m1 = mk<T>(v[0].x + C1 * t1.x + C2 * t2.x, v[0].y + C1 * t1.y + C2 * t2.y);
        const cx2<T> m2 = mk<T>(v[0].x + C2 * t1.x + C1 * t2.x, v[0].y + C2 * t1.y + C1 * t2.y);
        const cx2<T> n1 = mk<T>(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y), n2 = mk<T>(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
        v[0] = mk<T>(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
        if (SGN < 0) {      // y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2
            v[1] = mk<T>(m1.x + n1.y, m1.y - n1.x); v[4] = mk<T>(m1.x - n1.y, m1.y + n1.x);
            v[2] = mk<T>(m2.x + n2.y, m2.y - n2.x); v[3] = mk<T>(m2.x - n2.y, m2.y + n2.x);
        } else {
            v[1] = mk<T>(m1.x - n1.y, m1.y + n1.x); v[4] = mk<T>(m1.x + n1.y, m1.y - n1.x);
            v[2] = mk<T>(m2.x - n2.y, m2.y + n2.x); v[3] = mk<T>(m2.x + n2.y, m2.y - n2.x);
        }
    }
}

// The odd pass of a mixed-radix length M = RR s (s = 2^k), in front of the power-of-two passes: butterfly t takes elements
// t + q s, q < RR; output q' = DFT_RR(.)[q'] w_M^(t q') goes back to place q' and is element t of sub-transform q', a power-of-two
// transform of length s on the block [q' s, (q' + 1) s) -- what the passes below work through.  Twiddle indices t q' < M: no reduction.
template <int RR, typename T>
__device__ __forceinline__ void pass_odd_fwd(cx2<T> *__restrict__ S, int s, const cx2<T> *__restrict__ tw, int tid)
{
    for (int t = tid; t < s; t += FFT_THREADS) {
        cx2<T> v[RR];
#pragma unroll
        for (int q = 0; q < RR; ++q) v[q] = S[fft_pad(t + q * s)];
        dft_odd<RR, -1, T>(v);
#pragma unroll
        for (int q = 1; q < RR; ++q) v[q] = cmulf<T>(v[q], tw[t * q]);
#pragma unroll
        for (int q = 0; q < RR; ++q) S[fft_pad(t + q * s)] = v[q];
    }
}
template <int RR, typename T>
__device__ __forceinline__ void pass_odd_adj(cx2<T> *__restrict__ S, int s, const cx2<T> *__restrict__ tw, int tid)
{
    for (int t = tid; t < s; t += FFT_THREADS) {
        cx2<T> v[RR];
#pragma unroll
        for (int q = 0; q < RR; ++q) v[q] = S[fft_pad(t + q * s)];
#pragma unroll
        for (int q = 1; q < RR; ++q) v[q] = cmulcf<T>(v[q], tw[t * q]);
        dft_odd<RR, +1, T>(v);
#pragma unroll
        for (int q = 0; q < RR; ++q) S[fft_pad(t + q * s)] = v[q];
    }
}

// One in-place Gentleman-Sande pass over S[0..M): sub-transforms of length L = 2^lL split by radix R = 2^LR.
// Butterfly b = (blk, t): elements blk L + t + q (L/R); outputs y_q'[t] = DFT_R(.)[q'] w_L^(t q') stored at the same places.
// (nblk > 1: the nblk independent power-of-two transforms of length 2^logM a mixed-radix length falls into behind its odd pass, laid
// end to end: block index b >> ls runs over all of them, tw is the table of the power-of-two length)
template <int LR, typename T>
__device__ __forceinline__ void pass_fwd(cx2<T> *__restrict__ S, int logM, int lL, const cx2<T> *__restrict__ tw, int tid, int nblk = 1)
{
    constexpr int R = 1 << LR;
    const int M = 1 << logM, ls = lL - LR, s = 1 << ls, twsh = logM - lL;
    for (int b = tid; b < nblk * (M >> LR); b += FFT_THREADS) {
        const int blk = b >> ls, t = b & (s - 1);
        const int base = (blk << lL) + t;
        cx2<T> v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = S[fft_pad(base + (q << ls))];
        dft_small<R, -1, T>(v);
        cx2<T> w[R];
        twiddle_powers<R, T>(w, tw, t, twsh, M);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmulf<T>(v[q], w[q]);
#pragma unroll
        for (int q = 0; q < R; ++q) S[fft_pad(base + (q << ls))] = v[q];
    }
}
// its adjoint (the passes of the inverse transform, applied in reverse order)
template <int LR, typename T>
__device__ __forceinline__ void pass_adj(cx2<T> *__restrict__ S, int logM, int lL, const cx2<T> *__restrict__ tw, int tid, int nblk = 1)
{
    constexpr int R = 1 << LR;
    const int M = 1 << logM, ls = lL - LR, s = 1 << ls, twsh = logM - lL;
    for (int b = tid; b < nblk * (M >> LR); b += FFT_THREADS) {
        const int blk = b >> ls, t = b & (s - 1);
        const int base = (blk << lL) + t;
        cx2<T> v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = S[fft_pad(base + (q << ls))];
        cx2<T> w[R];
        twiddle_powers<R, T>(w, tw, t, twsh, M);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmulcf<T>(v[q], w[q]);
        dft_small<R, +1, T>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) S[fft_pad(base + (q << ls))] = v[q];
    }
}

// S holds a_j = x_j c_j at j = 1..n and zeros elsewhere (synchronised).  On return S[k] = sum_j a_j conj(c_{k-j}), k = 1..n
// (synchronised): forward passes, pointwise product with the chirp's transform, adjoint passes.
// all forward passes of a plan (synchronised behind each): the odd pass of a mixed-radix length first, then the power-of-two passes
// over its r blocks
template <typename T>
__device__ __forceinline__ void fft_forward(cx2<T> *__restrict__ S, const FftPlan<T> &P, int tid)
{
    if (P.r == 3) { pass_odd_fwd<3>(S, 1 << P.logM, P.tw, tid); __syncthreads(); }
    else if (P.r == 5) { pass_odd_fwd<5>(S, 1 << P.logM, P.tw, tid); __syncthreads(); }
    int lL = P.logM;
    for (int p = 0; p < P.npass; ++p) {
        const int lr = P.lr[p];
        if (lr == 4) pass_fwd<4>(S, P.logM, lL, P.tw2, tid, P.r);
        else if (lr == 3) pass_fwd<3>(S, P.logM, lL, P.tw2, tid, P.r);
        else if (lr == 2) pass_fwd<2>(S, P.logM, lL, P.tw2, tid, P.r);
        else pass_fwd<1>(S, P.logM, lL, P.tw2, tid, P.r);
        lL -= lr;
        __syncthreads();
    }
}

template <typename T>
__device__ __forceinline__ void chirp_convolve(cx2<T> *__restrict__ S, const FftPlan<T> &P, int tid)
{
    fft_forward<T>(S, P, tid);
    for (int i = tid; i < P.M; i += FFT_THREADS) S[fft_pad(i)] = cmulf<T>(S[fft_pad(i)], P.bhat[i]);
    __syncthreads();
    int lL = P.logM;
    for (int p = 0; p < P.npass; ++p) lL -= P.lr[p];
    for (int p = P.npass - 1; p >= 0; --p) {
        const int lr = P.lr[p];
        lL += lr;
        if (lr == 4) pass_adj<4>(S, P.logM, lL, P.tw2, tid, P.r);
        else if (lr == 3) pass_adj<3>(S, P.logM, lL, P.tw2, tid, P.r);
        else if (lr == 2) pass_adj<2>(S, P.logM, lL, P.tw2, tid, P.r);
        else pass_adj<1>(S, P.logM, lL, P.tw2, tid, P.r);
        __syncthreads();
    }
    if (P.r == 3) { pass_odd_adj<3>(S, 1 << P.logM, P.tw, tid); __syncthreads(); }
    else if (P.r == 5) { pass_odd_adj<5>(S, 1 << P.logM, P.tw, tid); __syncthreads(); }
}

// folded right-hand side g = lap - ring neighbours (seamlessClone_imp.cpp:1992-2008) at interior point (x, y), 0-based
__device__ __forceinline__ float fft_g(const Field &U, const Field &F, int c, int x, int y)
{
    const int w = U.W - 2, h = U.H - 2;
    const size_t o = (size_t)(y + 1) * U.pitch + (x + 1);
    const float *__restrict__ u = U.at(c);
    float v = F.at(c)[o];
    if (x == 0) v -= u[o - 1];
    if (y == 0) v -= u[o - U.pitch];
    if (x == w - 1) v -= u[o + 1];
    if (y == h - 1) v -= u[o + U.pitch];
    return v;
}

// MODE 0: rows of the folded right-hand side (from the fields) -> T[c][y][x]
// MODE 1: rows of `in` (the transposed plane: row = x, entries = y) -> DST, / den, DST -> out, same layout
// MODE 2: rows of `in` [c][y][x] -> DST, scale -> interior of the field U
template <int MODE, typename T>
__global__ __launch_bounds__(FFT_THREADS) void k_fft_dst(FftPlan<T> P, Field U, Field F, const T *__restrict__ in, T *__restrict__ out,
                                                         int rows, const float *__restrict__ f_row, const float *__restrict__ f_k, int exact,
                                                         double scale, int tstore)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_smem[];
    cx2<T> *__restrict__ S = reinterpret_cast<cx2<T> *>(fft_smem);
    const int tid = threadIdx.x, r = blockIdx.x, c = blockIdx.y, n = P.n;
    const T *__restrict__ src = (MODE == 0) ? nullptr : in + ((size_t)c * rows + r) * n;
    for (int i = tid; i < P.M; i += FFT_THREADS) {
        cx2<T> a = mk<T>((T)0, (T)0);
        if (i >= 1 && i <= n) {
            const T x = (MODE == 0) ? (T)fft_g(U, F, c, i - 1, r) : src[i - 1];
            const cx2<T> ch = P.chirp[i];
            a = mk<T>(x * ch.x, x * ch.y);
        }
        S[fft_pad(i)] = a;
    }
    __syncthreads();
    chirp_convolve<T>(S, P, tid);
    if (MODE == 1) {
        // X_k = Im(c_k y_k); divide by the reference's denominator (seamlessClone_imp.cpp:1651-1653: float tables added in float,
        // then - 4) and feed the quotient straight into the second transform: the row stays in LDS
        // (element k is read and rewritten by the same thread, the zeroed elements are read by nobody here: no barrier in between)
        for (int k = 1 + tid; k <= n; k += FFT_THREADS) {
            const cx2<T> y = S[fft_pad(k)], ch = P.chirp[k];
            const T X = ch.x * y.y + ch.y * y.x;
            T den;
            if (exact) den = (T)((2.0 * cospi((double)(r + 1) / (double)(rows + 1)) + 2.0 * cospi((double)k / (double)(n + 1))) - 4.0);
            else den = (T)((f_row[r] + f_k[k - 1]) - 4.0f);
            const T q = X / den;
            S[fft_pad(k)] = mk<T>(q * ch.x, q * ch.y);
        }
        for (int i = tid; i < P.M; i += FFT_THREADS) if (i == 0 || i > n) S[fft_pad(i)] = mk<T>((T)0, (T)0);
        __syncthreads();
        chirp_convolve<T>(S, P, tid);
    }
    for (int k = 1 + tid; k <= n; k += FFT_THREADS) {
        const cx2<T> y = S[fft_pad(k)], ch = P.chirp[k];
        const T X = ch.x * y.y + ch.y * y.x;
        if (MODE == 2) U.at(c)[(size_t)(r + 1) * U.pitch + k] = (float)(X * (T)scale);
        else if (tstore) out[((size_t)c * n + (k - 1)) * rows + r] = X;      // small planes: the transposed plane directly (see fft_solve_t)
        else out[((size_t)c * rows + r) * n + (k - 1)] = X;
    }
}

// out[c][x][y] = in[c][y][x]; 64 x 64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void k_fft_transpose(const T *__restrict__ in, T *__restrict__ out, int rows, int cols)
{
    __shared__ T t[64][65];
    const int c = blockIdx.z, x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const T *__restrict__ src = in + (size_t)c * rows * cols;
    T *__restrict__ dst = out + (size_t)c * rows * cols;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = y0 + ly + 4 * k, x = x0 + lx;
        t[ly + 4 * k][lx] = (y < rows && x < cols) ? src[(size_t)y * cols + x] : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int x = x0 + ly + 4 * k, y = y0 + lx;
        if (x < cols && y < rows) dst[(size_t)x * rows + y] = t[lx][ly + 4 * k];
    }
}

// ---------------------------------------------------------------------------------------------- host side
// The circular convolution's length for n unknowns: the shortest M = r 2^k >= 2n - 1 with r in {1, 3, 5} (round 5; powers of two only
// until then: 592 unknowns -> 2048 where 1280 does, 300 -> 1024 where 640 does).  The reference hands cuFFT 2n + 2 whatever it is
// (seamlessClone_imp.cpp:1694-1918).  Odd factors only in front of at least a 16-point power-of-two part.
struct FftLen { int r, logM; int M() const { return r << logM; } };
static FftLen fft_len(int n)
{
    const int need = std::max(2 * n - 1, 2);
    FftLen best{ 1, 1 };
    while (best.M() < need) ++best.logM;
    for (int r : { 3, 5 })
        for (int k = 4; (r << k) < best.M(); ++k)
            if ((r << k) >= need) { best = FftLen{ r, k }; break; }
    return best;
}

static void fft_radices(int logM, int lr[4], int &npass)
{
    npass = 0;
    int rem = logM;
    while (rem >= 4) { lr[npass++] = 4; rem -= 4; }
    if (rem > 0) lr[npass++] = rem;
}

// ---- the tables of one transform length, built on the device --------------------------------------------------------------
// (Round 3 built them on the host: a serial radix-2 double FFT with cos / sin in its inner loop plus M sincos for the
// twiddles, behind a stream synchronisation -- ~0.5 ms per direction at M = 2048, several times the 0.1-0.3 ms clone that
// needed them.  The reference builds its per-size tables with a kernel inside every call: seamlessClone_imp.cpp:569-603.)
// The tables of one transform length: chirp c_m = exp(i pi m^2 / N), m = 0 .. n (phase reduced exactly: m^2 mod 2N in integers), the
// twiddles exp(-2 pi i k / M), in double, stored as T (`tw64` additionally receives the twiddles in double when the transform of
// the chirp kernel runs in double while T is float), and the transform of the chirp kernel b (b_0 = 1, b_m = b_{M-m} = conj(c_m),
// m = 1 .. n-1, zero elsewhere) by the SAME in-LDS forward passes the solve uses -- their output order (digit reversed) is the order
// the pointwise product wants -- computed in TC (double whenever the row fits the LDS: M <= 8192) and stored as T with the 1/M of
// the inverse folded in.
// k_fft_build (round 4, late): the tables of up to TWO transform lengths (the two directions of a solve) in ONE launch, one workgroup
// per length -- until then two kernels per length (k_fft_tables, k_fft_bhat): a first call at a new ROI size spent four launches and two
// forks on them, and a 300 x 200 clone is nine launches in all, bound by the host's enqueue time.  Same arithmetic, value for value:
// the workgroup writes chirp and twiddles, then transforms the chirp kernel with the twiddles it has just written (visible to itself
// behind the barrier).
template <typename TC, typename T>
struct FftBuild { cx2<T> *chirp, *bhat, *tw, *tw2; cx2<double> *tw64, *tw64_2; FftPlan<TC> P; };      // P.tw / P.tw2: the twiddles the build's own transform reads (tw / tw2, or their double twins when TC is double and T float); tw2 == nullptr: a power-of-two length (tw serves)
template <typename TC, typename T>
struct FftBuildPair { FftBuild<TC, T> b[2]; };

template <typename TC, typename T>
__global__ __launch_bounds__(FFT_THREADS) void k_fft_build(FftBuildPair<TC, T> bp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_smem[];
    cx2<TC> *__restrict__ S = reinterpret_cast<cx2<TC> *>(fft_smem);
    const FftBuild<TC, T> &B = bp.b[blockIdx.x];
    const FftPlan<TC> &P = B.P;
    const int tid = threadIdx.x, n = P.n, M = P.M;
    const long long N2 = 4LL * (n + 1);
    for (int i = tid; i < max(M, n + 1); i += FFT_THREADS) {          // chirp and twiddles
        if (i <= n) {
            const long long q = ((long long)i * i) % N2;
            double sn, cs;
            sincospi((double)q / (double)(2 * (n + 1)), &sn, &cs);
            B.chirp[i] = mk<T>((T)cs, (T)sn);
        }
        if (i < M) {
            double sn, cs;
            sincospi(-2.0 * (double)i / (double)M, &sn, &cs);
            B.tw[i] = mk<T>((T)cs, (T)sn);
            if (B.tw64) B.tw64[i] = mk<double>(cs, sn);
        }
        if (B.tw2 && i < (1 << P.logM)) {                              // a mixed-radix length: the power-of-two passes' own table
            double sn, cs;
            sincospi(-2.0 * (double)i / (double)(1 << P.logM), &sn, &cs);
            B.tw2[i] = mk<T>((T)cs, (T)sn);
            if (B.tw64_2) B.tw64_2[i] = mk<double>(cs, sn);
        }
    }
    for (int i = tid; i < M; i += FFT_THREADS) {                      // the chirp kernel, then its transform
        const int m = i <= M / 2 ? i : M - i;
        cx2<TC> v = mk<TC>((TC)0, (TC)0);
        if (m == 0 && i == 0) v = mk<TC>((TC)1, (TC)0);
        else if (m >= 1 && m <= n - 1) {
            const long long q = ((long long)m * m) % N2;
            double sn, cs;
            sincospi((double)q / (double)(2 * (n + 1)), &sn, &cs);
            v = mk<TC>((TC)cs, (TC)(-sn));
        }
        S[fft_pad(i)] = v;
    }
    __threadfence();
    __syncthreads();
    fft_forward<TC>(S, P, tid);
    const TC inv = (TC)1 / (TC)M;
    for (int i = tid; i < M; i += FFT_THREADS) {
        const cx2<TC> v = S[fft_pad(i)];
        B.bhat[i] = mk<T>((T)(v.x * inv), (T)(v.y * inv));
    }
}

template <typename T>
static FftPlan<T> fft_plan_raw(const cx2<T> *chirp, int n, int logM)
{
    FftPlan<T> P{};
    P.chirp = chirp;
    P.bhat = chirp + (n + 1);
    P.tw = P.bhat + ((size_t)1 << logM);
    P.tw2 = P.tw;
    P.n = n; P.logM = logM; P.M = 1 << logM; P.r = 1;
    fft_radices(logM, P.lr, P.npass);
    return P;
}

// The tables for n unknowns in precision T: from the instance's LRU, or queued for a build on the second stream (fft_flush_builds:
// both directions of a solve in one launch; the caller makes the main stream wait for I->fft.ev_built before the first transform
// launch).  `keep`: an entry that must not be evicted (the other direction of the same solve).
template <typename T>
static int fft_build_dim(Instance *I, FftDim *&out, int n, const FftDim *keep)
{
    const bool dbl = sizeof(T) == sizeof(double);
    FftState &S = I->fft;
    FftDim *victim = nullptr;
    for (FftDim &d : S.dims) {
        if (d.chirp.p && d.n == n && d.dbl == dbl) { d.used = ++S.tick; out = &d; return SC_OK; }
        if (&d != keep && (!victim || d.used < victim->used)) victim = &d;
    }
    FftDim &D = *victim;
    I->info.new_size = 1;
    const FftLen L = fft_len(n);
    const int logM = L.logM, M = L.M();
    int rc;
    if (S.pending && !S.forked && S.nreq == 0) {     // a build of an EARLIER solve nobody waited for: ordered in front of whatever follows on the main stream
        SC_HIP(I, hipStreamWaitEvent(I->stream, S.ev_built, 0));      // (ensure() waits for that stream before it frees a buffer)
        S.pending = false;
    }
    const size_t bytes = sizeof(cx2<T>) * (4 * (size_t)M + 1);      // chirp[n + 1 <= M + 1] | bhat[M] | tw[M] | tw2[2^logM <= M]: sized by M alone, so an entry is reallocated only when M grows
    D.n = 0;
    if ((rc = ensure(I, D.chirp, bytes))) return rc;
    D.n = n; D.logM = logM; D.r = L.r; D.dbl = dbl; D.used = ++S.tick;
    S.req[S.nreq++] = &D;
    out = &D;
    return SC_OK;
}

// launches the queued builds (at most two: the directions of one solve) behind ONE fork of the second stream
template <typename T>
static int fft_flush_builds(Instance *I)
{
    FftState &S = I->fft;
    if (S.nreq == 0) return SC_OK;
    const bool dbl = sizeof(T) == sizeof(double);
    int rc;
    bool tcd[2] = { false, false };
    size_t maxM = 0;
    for (int k = 0; k < S.nreq; ++k) {
        tcd[k] = dbl || ((size_t)S.req[k]->r << S.req[k]->logM) <= ((size_t)1 << (FFT_MAX_LOGM - 1));      // the build's own transform in double whenever its row fits the LDS
        maxM = std::max(maxM, (size_t)S.req[k]->r << S.req[k]->logM);
    }
    // (per direction: tw64[M] | tw64_2[2^logM])
    if (!dbl && (tcd[0] || tcd[1]) && (rc = ensure(I, S.tw64, sizeof(cx2<double>) * 4 * maxM))) { for (int k = 0; k < S.nreq; ++k) S.req[k]->n = 0; S.nreq = 0; return rc; }
    if (!S.ev_fork) {
        SC_HIP(I, hipEventCreateWithFlags(&S.ev_fork, hipEventDisableTiming));
        SC_HIP(I, hipEventCreateWithFlags(&S.ev_built, hipEventDisableTiming));
    }
    // on the second stream, behind everything the main stream has been given so far (launches that still read the evicted tables);
    // ONE fork per solve: the eigenvalue tables follow on the same stream (every fork is an event record on the main stream, a
    // packet the next launch waits for)
    if (!S.forked) {
        SC_HIP(I, hipEventRecord(S.ev_fork, I->stream));
        SC_HIP(I, hipStreamWaitEvent(I->aux, S.ev_fork, 0));
        S.forked = true;
    }
    auto fill = [&](auto &B, int k) {
        using TC = typename std::remove_reference<decltype(B.P)>::type;
        (void)sizeof(TC);
        FftDim &D = *S.req[k];
        const int M = D.r << D.logM;
        cx2<T> *chirp = (cx2<T> *)D.chirp.p;
        B.chirp = chirp; B.bhat = chirp + (D.n + 1); B.tw = B.bhat + M;
        B.tw2 = D.r > 1 ? B.tw + M : nullptr;
        B.tw64 = (!dbl && tcd[k]) ? (cx2<double> *)S.tw64.p + (size_t)k * 2 * maxM : nullptr;
        B.tw64_2 = (B.tw64 && D.r > 1) ? B.tw64 + M : nullptr;
        B.P.chirp = nullptr; B.P.bhat = nullptr;
        B.P.n = D.n; B.P.logM = D.logM; B.P.M = M; B.P.r = D.r;
        fft_radices(D.logM, B.P.lr, B.P.npass);
    };
    // directions whose builds run in the same arithmetic share a launch
    for (int k = 0; k < S.nreq;) {
        const int cnt = (k + 1 < S.nreq && tcd[k + 1] == tcd[k]) ? 2 : 1;
        const size_t lds_elems = (size_t)fft_pad(std::max(S.req[k]->r << S.req[k]->logM, cnt > 1 ? S.req[k + 1]->r << S.req[k + 1]->logM : 0)) + 1;
        if (dbl) {
            if constexpr (sizeof(T) == sizeof(double)) {
                FftBuildPair<double, double> bp{};
                for (int q = 0; q < cnt; ++q) { fill(bp.b[q], k + q); bp.b[q].P.tw = bp.b[q].tw; bp.b[q].P.tw2 = bp.b[q].tw2 ? bp.b[q].tw2 : bp.b[q].tw; }
                hipLaunchKernelGGL((k_fft_build<double, double>), dim3(cnt), dim3(FFT_THREADS), sizeof(cx2<double>) * lds_elems, I->aux, bp);
            }
        } else if (tcd[k]) {
            if constexpr (sizeof(T) == sizeof(float)) {
                FftBuildPair<double, float> bp{};
                for (int q = 0; q < cnt; ++q) { fill(bp.b[q], k + q); bp.b[q].P.tw = bp.b[q].tw64; bp.b[q].P.tw2 = bp.b[q].tw64_2 ? bp.b[q].tw64_2 : bp.b[q].tw64; }
                hipLaunchKernelGGL((k_fft_build<double, float>), dim3(cnt), dim3(FFT_THREADS), sizeof(cx2<double>) * lds_elems, I->aux, bp);
            }
        } else {
            if constexpr (sizeof(T) == sizeof(float)) {
                FftBuildPair<float, float> bp{};
                for (int q = 0; q < cnt; ++q) { fill(bp.b[q], k + q); bp.b[q].P.tw = bp.b[q].tw; bp.b[q].P.tw2 = bp.b[q].tw2 ? bp.b[q].tw2 : bp.b[q].tw; }
                hipLaunchKernelGGL((k_fft_build<float, float>), dim3(cnt), dim3(FFT_THREADS), sizeof(cx2<float>) * lds_elems, I->aux, bp);
            }
        }
        k += cnt;
    }
    S.nreq = 0;
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipEventRecord(S.ev_built, I->aux));
    S.pending = true;
    return SC_OK;
}

// The reference's float tables fx[w] + fy[h] (seamlessClone_imp.cpp:596-599; PI is the float literal of seamlessClone_imp.h:17), computed on the
// host -- its libm is the one the CPU oracle's tables come from, and an entry that rounds the other way moves the lowest modes by
// grey levels -- into an LRU entry's OWN pinned staging: no stream synchronisation, the entry's upload event is waited for only
// when the entry is reused for another size (long complete by then).
static int fft_fxy(Instance *I, FftFxy *&out, int w, int h)
{
    FftState &S = I->fft;
    FftFxy *victim = nullptr;
    for (FftFxy &f : S.fxy) {
        if (f.d.p && f.w == w && f.h == h) { f.used = ++S.tick; out = &f; return SC_OK; }
        if (!victim || f.used < victim->used) victim = &f;
    }
    FftFxy &X = *victim;
    I->info.new_size = 1;
    int rc;
    if (X.ev) SC_HIP(I, hipEventSynchronize(X.ev));
    else SC_HIP(I, hipEventCreateWithFlags(&X.ev, hipEventDisableTiming));
    X.w = 0;
    if ((rc = ensure(I, X.d, sizeof(float) * (size_t)(w + h)))) return rc;
    // every entry's pinned staging is a piece of ONE block allocated with the first (w + h <= 16 384 floats: fft_supported): a
    // hipHostMalloc per entry made the first four new sizes of an instance 0.1 ms slower each (new_size leg, calls #0 .. #3)
    constexpr size_t FXY_STAGE = sizeof(float) * 16384;
    if ((rc = ensure_pinned(I, S.hst_all, FXY_STAGE * FftState::FXY))) return rc;
    float *fx = (float *)((unsigned char *)S.hst_all.p + FXY_STAGE * (size_t)(&X - S.fxy)), *fy = fx + w;
    const double PIf = (double)3.14159265358979323846f;
    for (int i = 0; i < w; ++i) fx[i] = (float)(2.0 * std::cos(PIf / (w + 1.0) * (i + 1.0)));
    for (int j = 0; j < h; ++j) fy[j] = (float)(2.0 * std::cos(PIf / (h + 1.0) * (j + 1.0)));
    X.singular = !((fx[0] + fy[0]) - 4.0f < 0.0f);
    // (on the main stream, behind the zero-fill ensure() gives a fresh buffer there: tried on the second stream beside the table
    // build, the upload raced that fill -- eigenvalue tables of zeros whenever the main stream was still busy)
    SC_HIP(I, hipMemcpyAsync(X.d.p, fx, sizeof(float) * (size_t)(w + h), hipMemcpyHostToDevice, I->stream));
    SC_HIP(I, hipEventRecord(X.ev, I->stream));
    X.w = w; X.h = h; X.used = ++S.tick;
    out = &X;
    return SC_OK;
}

template <typename T>
static FftPlan<T> fft_plan_of(const FftDim &D)
{
    FftPlan<T> P{};
    P.chirp = (const cx2<T> *)D.chirp.p;
    P.bhat = P.chirp + (D.n + 1);
    const size_t M = (size_t)D.r << D.logM;
    P.tw = P.bhat + M;
    P.tw2 = D.r > 1 ? P.tw + M : P.tw;
    P.n = D.n; P.logM = D.logM; P.M = (int)M; P.r = D.r;
    fft_radices(D.logM, P.lr, P.npass);
    return P;
}

// Opt the three transform kernels in to more than 64 KB of dynamic LDS.  The attribute belongs to the (function, DEVICE) pair:
// once per instance -- an instance lives on one device -- not once per process (a second GPU in the same process would never opt in).
template <typename T>
static hipError_t fft_opt_in_lds(Instance *I)
{
    bool &done = sizeof(T) == sizeof(double) ? I->fft_lds_double : I->fft_lds_float;
    if (done) return hipSuccess;
    const int max_log = sizeof(T) == sizeof(double) ? FFT_MAX_LOGM - 1 : FFT_MAX_LOGM;
    const int bytes = (int)(sizeof(cx2<T>) * (size_t)fft_pad(1 << max_log)) + 64;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_dst<0, T>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_dst<1, T>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_dst<2, T>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    // the table builders: a double row of up to 2^(FFT_MAX_LOGM - 1) points, or a float row of 2^FFT_MAX_LOGM
    const int bbytes = (int)(sizeof(cx2<double>) * (size_t)fft_pad(1 << (FFT_MAX_LOGM - 1))) + 64;
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_build<double, T>), hipFuncAttributeMaxDynamicSharedMemorySize, bbytes);
    if (e == hipSuccess && sizeof(T) == sizeof(float))
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_build<float, float>), hipFuncAttributeMaxDynamicSharedMemorySize, bbytes);
    done = e == hipSuccess;
    return e;
}

bool fft_supported(int w, int h, bool fp64)
{
    const int mx = 1 << (fp64 ? FFT_MAX_LOGM - 1 : FFT_MAX_LOGM);        // double: 16 bytes per element, n <= 4096 per side
    return w >= 1 && h >= 1 && fft_len(w).M() <= mx && fft_len(h).M() <= mx;
}

template <typename T>
static int fft_solve_t(Instance *I)
{
    const int w = I->F.W - 2, h = I->F.H - 2, C = I->F.C;
    SC_HIP(I, fft_opt_in_lds<T>(I));
    FftState &S = I->fft;
    int rc;
    FftDim *dw = nullptr, *dh = nullptr;
    FftFxy *X = nullptr;
    S.forked = false;
    S.nreq = 0;
    if ((rc = fft_build_dim<T>(I, dw, w, nullptr))) return rc;
    if ((rc = fft_build_dim<T>(I, dh, h, dw))) { for (int k = 0; k < S.nreq; ++k) S.req[k]->n = 0; S.nreq = 0; return rc; }      // (queued entries hold no tables yet)
    if ((rc = fft_flush_builds<T>(I))) return rc;
    const size_t plane = (size_t)w * h;
    if ((rc = ensure(I, S.A, sizeof(T) * plane * C, false))) return rc;      // (every launch below writes the whole plane it hands on)
    if ((rc = ensure(I, S.B, sizeof(T) * plane * C, false))) return rc;
    if ((rc = fft_fxy(I, X, w, h))) return rc;
    if (S.pending) {                  // tables of a new length are being built on the second stream
        SC_HIP(I, hipStreamWaitEvent(I->stream, S.ev_built, 0));
        S.pending = false;
    }
    const int exact = (X->singular || (I->opts.flags & SC_FLAG_EXACT_TABLES)) ? 1 : 0;
    const FftPlan<T> Pw = fft_plan_of<T>(*dw), Ph = fft_plan_of<T>(*dh);
    const size_t ldsw = sizeof(cx2<T>) * (size_t)(fft_pad(Pw.M) + 1), ldsh = sizeof(cx2<T>) * (size_t)(fft_pad(Ph.M) + 1);
    Field &U = I->result_in_U1 ? I->U1 : I->U0;
    T *A = (T *)S.A.p, *B = (T *)S.B.p;
    const float *fx = (const float *)X->d.p, *fy = fx + w;
    const double scale = 4.0 / ((w + 1.0) * (h + 1.0));
    const dim3 tg_hw((w + 63) / 64, (h + 63) / 64, C), tg_wh((h + 63) / 64, (w + 63) / 64, C);
    // Small planes (cache resident: the scattered 4- or 8-byte stores of a transposed write cost nothing there) skip the two
    // transpose launches: each transform launch writes the plane the next one reads row-wise.  Three launches instead of five:
    // 0.057 -> ~0.04 ms of a 298 x 192 clone's solve, where every launch is at its latency floor.
    const bool tiny = plane * sizeof(T) <= ((size_t)4 << 20);
    if (tiny) {
        hipLaunchKernelGGL((k_fft_dst<0, T>), dim3(h, C), dim3(FFT_THREADS), ldsw, I->stream, Pw, U, I->F, (const T *)nullptr, B, h, fx, fy, exact, 1.0, 1);   // B[c][x][y]
        hipLaunchKernelGGL((k_fft_dst<1, T>), dim3(w, C), dim3(FFT_THREADS), ldsh, I->stream, Ph, U, I->F, (const T *)B, A, w, fx, fy, exact, 1.0, 1);        // A[c][y][x]
        hipLaunchKernelGGL((k_fft_dst<2, T>), dim3(h, C), dim3(FFT_THREADS), ldsw, I->stream, Pw, U, I->F, (const T *)A, B, h, fx, fy, exact, scale, 0);
    } else {
    hipLaunchKernelGGL((k_fft_dst<0, T>), dim3(h, C), dim3(FFT_THREADS), ldsw, I->stream, Pw, U, I->F, (const T *)nullptr, A, h, fx, fy, exact, 1.0, 0);
    hipLaunchKernelGGL((k_fft_transpose<T>), tg_hw, dim3(256), 0, I->stream, (const T *)A, B, h, w);                 // B[c][x][y]
    hipLaunchKernelGGL((k_fft_dst<1, T>), dim3(w, C), dim3(FFT_THREADS), ldsh, I->stream, Ph, U, I->F, (const T *)B, A, w, fx, fy, exact, 1.0, 0);
    hipLaunchKernelGGL((k_fft_transpose<T>), tg_wh, dim3(256), 0, I->stream, (const T *)A, B, w, h);                 // B[c][y][x]
    hipLaunchKernelGGL((k_fft_dst<2, T>), dim3(h, C), dim3(FFT_THREADS), ldsw, I->stream, Pw, U, I->F, (const T *)B, A, h, fx, fy, exact, scale, 0);
    }
    SC_HIP(I, hipGetLastError());
    I->info.sweeps = 1;
    I->info.converged = 1;
    I->info.sweep_launches += 3;
    return SC_OK;
}

// Direct solve of the fields bound to the instance: interior of result(I) <- the reference's answer.  F must be float.
int fft_solve(Instance *I, bool fp64)
{
    if (I->f_half) { I->err = "internal: float16 right-hand side on the direct path"; return SC_ERR_BAD_ARG; }
    const int w = I->F.W - 2, h = I->F.H - 2;
    if (!fft_supported(w, h, fp64)) {
        I->err = fp64 ? "SC_METHOD_FFT with SC_FLAG_FFT_FP64: more than 4096 unknowns per side" : "SC_METHOD_FFT: more than 8192 unknowns per side";
        return SC_ERR_BAD_SIZE;
    }
    return fp64 ? fft_solve_t<double>(I) : fft_solve_t<float>(I);
}

} // namespace sc
