// sc_fdmid.hip -- direct solve of ONE MID level of the multigrid hierarchy on the fp32 matrix cores.
//
// A single clone spends a third of every V-cycle in the levels below level 2: for a 2048^2 ROI the 255^2, 127^2 and <= 63^2 levels
// cost 5.4 + 5.4 + 16.3 + 6.1 + 5.7 = 39 us per cycle in five dependent, latency-bound launches for 1.5 % of the unknowns.  Every
// level operator is a tensor sum Tx (x) I + I (x) Ty of two tridiagonal 1-D operators (the irregular last interval only changes
// their last row), so with T = V L V^-1 the level is solved EXACTLY by
//     U = Vy [ (Vy^-1 F Vx^-T) / (ly_j + lx_i) ] Vx^T
// -- the fast diagonalisation the LDS-resident bottom kernel already uses at <= 63^2 (sc_multigrid.cpp, build_fd), here as four
// dense products of <= 384^2 matrices per channel on v_mfma_f32_16x16x4_f32: four launches instead of five, none of them
// waiting on a 1024-thread workgroup.  float32 is enough: the level solves for a CORRECTION, and level 0's residual decides what
// the cycles converge to.  Used for single clones (few channels); groups keep the V-cycle (their coarse launches are
// bandwidth bound, and 48 channels of dense products would cost more than they save).
#include "sc_instance.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace sc {

typedef float v4f32 __attribute__((ext_vector_type(4)));

constexpr int SG_BM = 128, SG_BN = 128, SG_BK = 16, SG_PAD = 4;

// C = A * B, row-major, float, one 128 x 128 tile per workgroup (8 waves, each 32 x 64 = 2 x 4 MFMA tiles), blockIdx.z = channel.
// M, N are covered by the grid in whole tiles, K in steps of 16; rows of A at or beyond a_rows and its columns at or beyond
// a_cols count as zero (their addresses are clamped), B is fully padded.  Strides per channel may be 0 (a shared matrix).
// EPI 0: plain store into a padded plane.  EPI 1: multiply by dinv[row][col] (ldd) first.
// EPI 2: store only rows < c_rows, columns < c_cols, at C + row * ldc + col (the interior of a level field: the caller passes
// the address of interior point (0, 0)).
template <int EPI>
__global__ __launch_bounds__(512) void k_sgemm(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                               int lda, int ldb, int ldc, int K, size_t strideA, size_t strideB, size_t strideC,
                                               int a_rows, int a_cols, const float *__restrict__ dinv, int ldd, int c_rows, int c_cols)
{
    __shared__ float As[2][SG_BK][SG_BM + SG_PAD];
    __shared__ float Bs[2][SG_BK][SG_BN + SG_PAD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * SG_BM, n0 = blockIdx.x * SG_BN, zc = blockIdx.z;
    A += (size_t)zc * strideA; B += (size_t)zc * strideB; C += (size_t)zc * strideC;
    const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 5, bc = (t & 31) * 4;
    const bool arow_ok = (m0 + ar) < a_rows;
    const float *__restrict__ ap = A + (size_t)min(m0 + ar, a_rows - 1) * lda;
    const float *__restrict__ bp = B + (size_t)bk * ldb + n0 + bc;
    float4 ra, rb;
    auto load = [&](int k0) {
        const int kc = k0 + ak;                                   // 4 consecutive k of one row of A
        ra = *reinterpret_cast<const float4 *>(ap + (kc < a_cols ? kc : 0));      // (a group that starts inside the row may run up to 3 floats past a_cols: inside the row pitch)
        if (!arow_ok || kc >= a_cols) ra = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (kc + 3 >= a_cols) {                              // the row ends inside this group (a_cols is not a multiple of 4)
            if (kc + 1 >= a_cols) ra.y = 0.f;
            if (kc + 2 >= a_cols) ra.z = 0.f;
            ra.w = 0.f;
        }
        rb = *reinterpret_cast<const float4 *>(bp + (size_t)k0 * ldb);
    };
    auto store = [&](int s) {
        As[s][ak + 0][ar] = ra.x; As[s][ak + 1][ar] = ra.y; As[s][ak + 2][ar] = ra.z; As[s][ak + 3][ar] = ra.w;
        *reinterpret_cast<float4 *>(&Bs[s][bk][bc]) = rb;
    };
    v4f32 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f32){ 0.f, 0.f, 0.f, 0.f };
    const int fr = lane & 15, fk = lane >> 4;
    load(0);
    store(0);
    __syncthreads();
    const int ntiles = K / SG_BK;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int s = kt & 1;
        load(min(kt + 1, ntiles - 1) * SG_BK);
#pragma unroll
        for (int kk = 0; kk < SG_BK; kk += 4) {
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[s][kk + fk][wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[s][kk + fk][wn * 64 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        store(s ^ 1);
        __syncthreads();
    }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + 4 * fk + r, col = n0 + wn * 64 + j * 16 + fr;
                float v = acc[i][j][r];
                if (EPI == 1) v *= dinv[(size_t)row * ldd + col];
                if (EPI == 2) { if (row < c_rows && col < c_cols) C[(size_t)row * ldc + col] = v; }
                else C[(size_t)row * ldc + col] = v;
            }
}

template <int EPI>
static void launch_sgemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, size_t sA, size_t sB,
                         size_t sC, int a_rows, int a_cols, const float *dinv, int ldd, int c_rows, int c_cols, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_sgemm<EPI>, dim3(N / SG_BN, M / SG_BM, batch), dim3(512), 0, s, A, B, C, lda, ldb, ldc, K, sA, sB, sC, a_rows, a_cols,
                       dinv, ldd, c_rows, c_cols);
}

// the four products of one level: F = RHS field of the level (ring included), U = its correction field.
// Matrices (device, zero padded, row-major): R1[Pk][Px] (row 0 = the ring column of F, then Vx^-T), L1[Py][Py] = Vy^-1,
// R2[Px][Px] = Vx^T, L2[Py][Py] = Vy, Dinv[Py][Px]; G1, G2: scratch planes [C][Py][Px].
int fdmid_solve(Instance *I, const Field &F, const Field &U)
{
    FdMid &M = I->fdm;
    const int C = F.C, Px = M.Px, Py = M.Py, Pk = M.Pk;
    const size_t pl = (size_t)Py * Px;
    float *G1 = (float *)M.G1.p, *G2 = (float *)M.G2.p;
    const float *R1 = (const float *)M.mats.p, *L1 = R1 + (size_t)Pk * Px, *R2 = L1 + (size_t)Py * Py, *L2 = R2 + (size_t)Px * Px,
                *Dinv = L2 + (size_t)Py * Py;
    // G1 = F(interior rows, all columns from the ring on) R1
    launch_sgemm<0>(F.p + F.pitch, R1, G1, Py, Px, Pk, F.pitch, Px, Px, F.plane, 0, pl, M.ny, std::min(F.pitch, M.nx + 2), nullptr, 0, 0, 0, C, I->stream);
    // G2 = (L1 G1) .* Dinv
    launch_sgemm<1>(L1, G1, G2, Py, Px, Py, Py, Px, Px, 0, pl, pl, Py, Py, Dinv, Px, 0, 0, C, I->stream);
    // G1 = G2 R2
    launch_sgemm<0>(G2, R2, G1, Py, Px, Px, Px, Px, Px, pl, 0, pl, Py, Px, nullptr, 0, 0, 0, C, I->stream);
    // U(interior) = L2 G1
    launch_sgemm<2>(L2, G1, U.p + U.pitch + 1, Py, Px, Py, Py, Px, U.pitch, 0, pl, U.plane, Py, Py, nullptr, 0, M.ny, M.nx, C, I->stream);
    SC_HIP(I, hipGetLastError());
    return SC_OK;
}

// GPU self test of k_sgemm against a host triple loop (ragged A, shared / per-channel operands, the three epilogues).
// Returns the largest absolute error relative to the largest |C| (float rounding: < 1e-5), or a negative error code.
double fdmid_selftest(Instance *I)
{
    const int M = 256, N = 128, K = 48, a_rows = 201, a_cols = 37, batch = 3, lda = 64;
    std::vector<float> A((size_t)batch * a_rows * lda), B((size_t)K * N), D((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (auto &v : D) v = 0.5f + 0.25f * rnd();
    float *dA = nullptr, *dB = nullptr, *dC = nullptr, *dD = nullptr;
    const size_t cbytes = sizeof(float) * (size_t)batch * M * N;
    if (hipMalloc((void **)&dA, A.size() * 4 + 4096) != hipSuccess || hipMalloc((void **)&dB, B.size() * 4) != hipSuccess ||
        hipMalloc((void **)&dC, cbytes) != hipSuccess || hipMalloc((void **)&dD, D.size() * 4) != hipSuccess) return -1.0;
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dD, D.data(), D.size() * 4, hipMemcpyHostToDevice);
    double worst = 0.0;
    std::vector<float> Cg((size_t)batch * M * N);
    for (int epi = 0; epi < 3; ++epi) {
        (void)hipMemset(dC, 0, cbytes);
        const int c_rows = 190, c_cols = 77;
        if (epi == 0) launch_sgemm<0>(dA, dB, dC, M, N, K, lda, N, N, (size_t)a_rows * lda, 0, (size_t)M * N, a_rows, a_cols, nullptr, 0, 0, 0, batch, I->stream);
        if (epi == 1) launch_sgemm<1>(dA, dB, dC, M, N, K, lda, N, N, (size_t)a_rows * lda, 0, (size_t)M * N, a_rows, a_cols, dD, N, 0, 0, batch, I->stream);
        if (epi == 2) launch_sgemm<2>(dA, dB, dC, M, N, K, lda, N, N, (size_t)a_rows * lda, 0, (size_t)M * N, a_rows, a_cols, nullptr, 0, c_rows, c_cols, batch, I->stream);
        if (hipStreamSynchronize(I->stream) != hipSuccess) return -2.0;
        (void)hipMemcpy(Cg.data(), dC, cbytes, hipMemcpyDeviceToHost);
        double big = 1e-30, err = 0.0;
        for (int z = 0; z < batch; ++z)
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double want = 0.0;
                    if (m < a_rows)
                        for (int k = 0; k < std::min(K, a_cols); ++k) want += (double)A[((size_t)z * a_rows + m) * lda + k] * B[(size_t)k * N + n];
                    if (epi == 1) want *= D[(size_t)m * N + n];
                    if (epi == 2 && !(m < c_rows && n < c_cols)) want = 0.0;
                    big = std::max(big, std::fabs(want));
                    err = std::max(err, std::fabs(want - (double)Cg[((size_t)z * M + m) * N + n]));
                }
        worst = std::max(worst, err / big);
    }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
    return worst;
}

} // namespace sc
