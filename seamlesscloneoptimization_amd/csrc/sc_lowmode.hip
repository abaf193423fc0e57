// sc_lowmode.hip -- the float-table correction: turns the exact solution of the 5-point system into the answer
// OpenCV and the reference compute.
//
// What the reference does (seamlessClone_imp.cpp:1814-1896): u = DST^-1( DST(g) / den ), den[j][i] = filter_X[i] +
// filter_Y[j] - 4 with filter_X[i] = 2 cos(PI/(w+1) (i+1)) evaluated in double but STORED as float (:581-599; PI is the
// float literal of seamlessClone_imp.h:17) and the sum formed in float (:1651-1653).  For the lowest modes the true
// denominator is ~2 (pi/n)^2 ~ 5e-6 at n = 2048 while the float sum is only good to ~2e-7, so their amplitudes come out
// a few percent off: the reference's answer is up to 3 grey levels (2048^2) or 7 (4096^2) away from the exact
// solution of the system -- which is what the multigrid solver converges to.  Only the low modes are affected:
//     u_ref - u_exact = S^-1[ S(u_exact) * (den_exact / den_float - 1) ],        S = 2-D DST-I,
// and the factor is below 1e-4 beyond the first ~n/64 modes per direction.  So the correction is three small kernels
// on K = Kx x Ky modes (K ~ n/64 per direction; measured residual vs all modes: < 0.002 grey levels):
//   k_lm_project : T[k][x]  = sum_y Sy[y][k] U[y][x]                  (per row chunk; a second stage adds the chunks)
//   k_lm_coeffs  : Uh[k][l] = sum_x T[k][x] Sx[x][l];  E[k][x] = sum_l Uh[k][l] R[k][l] Sx[x][l]
//   k_lm_expand  : Out[y][x] = U[y][x] + sum_k Sy[y][k] E[k][x]
// Sy / Sx = sin tables (built on the device in double, sinpi with exact integer argument reduction), R = (den_exact /
// den_float - 1) * 4/((w+1)(h+1)) built on the host in double with the reference's float expressions.
// All sums run in a fixed order (no atomics): results are reproducible bit for bit, alone or in a group.
// Accuracy needed: the correction is a few grey levels and must be good to ~1e-3 of that, so float32 is ample.
#include "sc_instance.h"
#include <cmath>
#include <vector>
#include <cstring>
#include <algorithm>

namespace sc {

constexpr int LM_KB = 32;          // modes per register block
constexpr int LM_ROWS = 128;       // rows per projection chunk (4 waves x 32 rows)

int lowmode_count(int n)
{
    int k = (n + 63) / 64;
    k = (std::max(k, 8) + 7) & ~7;
    return std::min(std::min(k, n), 256);          // 256: k_lm_coeffs keeps one coefficient row in LDS (n <= 16384 is unaffected)
}

// S[i][l] = sin(pi (i+1)(l+1) / (n+1)), i < n, l < K; row pitch Kp floats, pad columns zero
__global__ __launch_bounds__(256) void k_lm_table(float *__restrict__ S, int n, int K, int Kp)
{
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)n * Kp) return;
    const int i = (int)(id / Kp), l = (int)(id % Kp);
    float v = 0.f;
    if (l < K) {
        const long m = 2L * (n + 1);
        const long r = ((long)(i + 1) * (l + 1)) % m;           // sin(pi r / (n+1)) has period 2(n+1) in r
        v = (float)sinpi((double)r / (double)(n + 1));
    }
    S[id] = v;
}

// T chunk: P[chunk][c][k][x] = sum over the chunk's rows y of Sy[y-1][k] * U[c][y][x]   (field coordinates: interior
// rows 1..H-2, every column of the pitch; ring / pad columns produce values nobody reads).
// Workgroup = 4 waves x 64 columns; wave v takes rows r0 + v, r0 + v + 4, ...; lane = one column; the Sy row is
// wave-uniform (scalar loads), the accumulators are LM_KB registers; the 4 waves are added through LDS in a fixed order.
__global__ __launch_bounds__(256) void k_lm_project(Field U, const float *__restrict__ Sy, int Kyp, int nkb, float *__restrict__ P)
{
    __shared__ float red[4][LM_KB][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.z, chunk = blockIdx.y;
    const int x = blockIdx.x * 64 + lane;                        // < pitch (grid covers the pitch exactly: pitch % 64 == 0)
    const int y0 = 1 + chunk * LM_ROWS, y1 = min(U.H - 1, y0 + LM_ROWS);
    const float *__restrict__ u = U.at(c) + x;
    const size_t P_ = U.pitch;
    for (int kb = 0; kb < nkb; ++kb) {
        float acc[LM_KB];
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) acc[k] = 0.f;
        for (int y = y0 + wv; y < y1; y += 4) {
            const float v = u[(size_t)y * P_];
            const float *__restrict__ s = Sy + (size_t)(y - 1) * Kyp + kb * LM_KB;
#pragma unroll
            for (int k = 0; k < LM_KB; ++k) acc[k] = __builtin_fmaf(s[k], v, acc[k]);
        }
        if (kb) __syncthreads();
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) red[wv][k][lane] = acc[k];
        __syncthreads();
        float *__restrict__ out = P + (((size_t)chunk * U.C + c) * Kyp + (size_t)kb * LM_KB) * P_ + blockIdx.x * 64;
        for (int i = threadIdx.x; i < LM_KB * 64; i += 256) {
            const int k = i >> 6, xx = i & 63;
            out[(size_t)k * P_ + xx] = ((red[0][k][xx] + red[1][k][xx]) + (red[2][k][xx] + red[3][k][xx]));
        }
    }
}

// One workgroup per (mode row k, channel c):  T[x] = sum_chunks P;  Uh[l] = sum_x T[x] Sx[x-1][l];
// E[c][k][x] = sum_l Uh[l] R[k][l] Sx[x-1][l].  Everything in a fixed order.
__global__ __launch_bounds__(256) void k_lm_coeffs(const float *__restrict__ P, int nchunks, int C, int W, int pitch, int Ky, int Kyp,
                                                    const float *__restrict__ Sx, int Kx, int Kxp,
                                                    const float *__restrict__ R, float *__restrict__ E)
{
    __shared__ float part[LM_KB][257];
    __shared__ float chat[256];                                  // Kx <= 256
    const int k = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    float *__restrict__ e = E + ((size_t)c * Kyp + k) * pitch;
    if (k >= Ky) {                                               // pad rows of the register blocks: no mode, no contribution
        for (int x = t; x < pitch; x += 256) e[x] = 0.f;
        return;
    }
    // T row (chunks added in order), parked in E's own row: each thread only ever touches its own columns
    for (int x = 1 + t; x <= W - 2; x += 256) {
        float s = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) s += P[(((size_t)ch * C + c) * Kyp + k) * pitch + x];
        e[x] = s;
    }
    for (int lb = 0; lb < Kx; lb += LM_KB) {
        float acc[LM_KB];
#pragma unroll
        for (int j = 0; j < LM_KB; ++j) acc[j] = 0.f;
        for (int x = 1 + t; x <= W - 2; x += 256) {
            const float tv = e[x];
            const float4 *__restrict__ row = reinterpret_cast<const float4 *>(Sx + (size_t)(x - 1) * Kxp + lb);
#pragma unroll
            for (int j = 0; j < LM_KB / 4; ++j) {
                const float4 s4 = row[j];
                acc[4 * j + 0] = __builtin_fmaf(tv, s4.x, acc[4 * j + 0]);
                acc[4 * j + 1] = __builtin_fmaf(tv, s4.y, acc[4 * j + 1]);
                acc[4 * j + 2] = __builtin_fmaf(tv, s4.z, acc[4 * j + 2]);
                acc[4 * j + 3] = __builtin_fmaf(tv, s4.w, acc[4 * j + 3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LM_KB; ++j) part[j][t] = acc[j];
        __syncthreads();
        // 256 partials per mode: thread t adds 32 of them for mode t/8, then the 8 lanes of a mode are added in a fixed tree
        {
            const int j = t >> 3, seg = t & 7;
            float s = 0.f;
            for (int i = 0; i < 32; ++i) s += part[j][seg * 32 + i];
            s += __shfl_down(s, 4, 8);
            s += __shfl_down(s, 2, 8);
            s += __shfl_down(s, 1, 8);
            if (seg == 0 && lb + j < Kx) chat[lb + j] = s * R[(size_t)k * Kxp + lb + j];
        }
    }
    __syncthreads();
    for (int x = 1 + t; x <= W - 2; x += 256) {
        const float *__restrict__ row = Sx + (size_t)(x - 1) * Kxp;
        float s = 0.f;
        for (int l = 0; l < Kx; ++l) s = __builtin_fmaf(chat[l], row[l], s);
        e[x] = s;
    }
}

// Out[c][y][x] = U[c][y][x] + sum_k Sy[y-1][k] E[c][k][x] on the interior; same tiling as the projection (lane = column,
// the E column lives in registers, Sy rows are scalar loads).
__global__ __launch_bounds__(256) void k_lm_expand(Field U, Field Out, const float *__restrict__ Sy, int Kyp, int nkb,
                                                   const float *__restrict__ E)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.z;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = 1 + blockIdx.y * LM_ROWS, y1 = min(U.H - 1, y0 + LM_ROWS);
    const size_t P_ = U.pitch;
    const float *__restrict__ u = U.at(c) + x;
    float *__restrict__ o = Out.at(c) + x;
    const bool inside = x >= 1 && x <= U.W - 2;
    float corr[LM_ROWS / 4];
#pragma unroll
    for (int i = 0; i < LM_ROWS / 4; ++i) corr[i] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        float e[LM_KB];
        const float *__restrict__ ep = E + ((size_t)c * Kyp + (size_t)kb * LM_KB) * P_ + x;
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) e[k] = ep[(size_t)k * P_];
#pragma unroll
        for (int i = 0; i < LM_ROWS / 4; ++i) {
            const int y = y0 + wv + 4 * i;
            if (y < y1) {
                const float *__restrict__ s = Sy + (size_t)(y - 1) * Kyp + kb * LM_KB;
                float a = corr[i];
#pragma unroll
                for (int k = 0; k < LM_KB; ++k) a = __builtin_fmaf(s[k], e[k], a);
                corr[i] = a;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LM_ROWS / 4; ++i) {
        const int y = y0 + wv + 4 * i;
        if (y < y1 && inside) o[(size_t)y * P_] = u[(size_t)y * P_] + corr[i];
    }
}

// ---------------------------------------------------------------------------------------------------------- host side

// R[k][l] = (den_exact / den_float - 1) * 4 / ((w+1)(h+1)).  den_float follows the reference to the letter:
// filter_X[i] = (float)(2.0 * cos(PI/(n+1.0) * (i+1.0))) with the FLOAT literal PI (seamlessClone_imp.h:17, .cpp:596-599),
// den = filter_X[i] + filter_Y[j] - 4 in float (:1651-1653).  den_exact = -4 (sin^2(a/2) + sin^2(b/2)) in double.
static void build_ratio(int w, int h, int Kx, int Ky, int Kxp, float *R)
{
    const double PIf = (double)3.14159265358979323846f;
    std::vector<float> fx(Kx), fy(Ky);
    for (int i = 0; i < Kx; ++i) fx[i] = (float)(2.0 * std::cos(PIf / (w + 1.0) * (i + 1.0)));
    for (int j = 0; j < Ky; ++j) fy[j] = (float)(2.0 * std::cos(PIf / (h + 1.0) * (j + 1.0)));
    const double scale = 4.0 / ((w + 1.0) * (h + 1.0));
    for (int j = 0; j < Ky; ++j)
        for (int i = 0; i < Kxp; ++i) {
            if (i >= Kx) { R[(size_t)j * Kxp + i] = 0.f; continue; }
            const double sa = std::sin(0.5 * M_PI * (i + 1.0) / (w + 1.0)), sb = std::sin(0.5 * M_PI * (j + 1.0) / (h + 1.0));
            const double den_e = -4.0 * (sa * sa + sb * sb);
            const float den_f = (fx[i] + fy[j]) - 4.0f;
            R[(size_t)j * Kxp + i] = (float)((den_e / (double)den_f - 1.0) * scale);
        }
}

// (Re)builds the tables for the fields currently bound to the instance; no-op when the geometry is unchanged.
static int lm_prepare(Instance *I)
{
    LowMode &L = I->lm;
    const int W = I->F.W, H = I->F.H, C = I->F.C, pitch = I->F.pitch;
    const int w = W - 2, h = H - 2;
    const int Kx = lowmode_count(w), Ky = lowmode_count(h);
    const int Kxp = round_up(Kx, LM_KB), Kyp = round_up(Ky, LM_KB);
    const int nchunks = (h + LM_ROWS - 1) / LM_ROWS;
    int rc;
    if ((rc = ensure(I, L.P, sizeof(float) * (size_t)nchunks * C * Kyp * pitch))) return rc;
    if ((rc = ensure(I, L.E, sizeof(float) * (size_t)C * Kyp * pitch))) return rc;
    L.C = C;
    if (L.w == w && L.h == h && L.Sx.p && L.Sy.p && L.R.p) return SC_OK;
    if ((rc = ensure(I, L.Sx, sizeof(float) * (size_t)w * Kxp))) return rc;
    if ((rc = ensure(I, L.Sy, sizeof(float) * (size_t)h * Kyp))) return rc;
    if ((rc = ensure(I, L.R, sizeof(float) * (size_t)Kyp * Kxp))) return rc;
    if ((rc = ensure_pinned(I, L.hR, sizeof(float) * (size_t)Kyp * Kxp))) return rc;
    std::memset(L.hR.p, 0, sizeof(float) * (size_t)Kyp * Kxp);
    build_ratio(w, h, Kx, Ky, Kxp, (float *)L.hR.p);
    SC_HIP(I, hipMemcpyAsync(L.R.p, L.hR.p, sizeof(float) * (size_t)Kyp * Kxp, hipMemcpyHostToDevice, I->stream));
    hipLaunchKernelGGL(k_lm_table, dim3((unsigned)(((size_t)w * Kxp + 255) / 256)), dim3(256), 0, I->stream, (float *)L.Sx.p, w, Kx, Kxp);
    hipLaunchKernelGGL(k_lm_table, dim3((unsigned)(((size_t)h * Kyp + 255) / 256)), dim3(256), 0, I->stream, (float *)L.Sy.p, h, Ky, Kyp);
    SC_HIP(I, hipGetLastError());
    L.w = w; L.h = h; L.Kx = Kx; L.Ky = Ky; L.Kxp = Kxp; L.Kyp = Kyp;
    return SC_OK;
}

// Out = U + correction (interior; ring and pads of Out are left as they are).  U and Out are fields of the instance's
// current shape; Out may not alias U.
int lowmode_correct(Instance *I, const Field &U, const Field &Out)
{
    if (U.W < 3 || U.H < 3) return SC_OK;
    int rc = lm_prepare(I);
    if (rc) return rc;
    const LowMode &L = I->lm;
    const int nchunks = (L.h + LM_ROWS - 1) / LM_ROWS, nkb = L.Kyp / LM_KB;
    const dim3 grid(U.pitch / 64, nchunks, U.C);
    hipLaunchKernelGGL(k_lm_project, grid, dim3(256), 0, I->stream, U, (const float *)L.Sy.p, L.Kyp, nkb, (float *)L.P.p);
    hipLaunchKernelGGL(k_lm_coeffs, dim3(L.Kyp, U.C), dim3(256), 0, I->stream, (const float *)L.P.p, nchunks, U.C, U.W, U.pitch,
                       L.Ky, L.Kyp, (const float *)L.Sx.p, L.Kx, L.Kxp, (const float *)L.R.p, (float *)L.E.p);
    hipLaunchKernelGGL(k_lm_expand, grid, dim3(256), 0, I->stream, U, Out, (const float *)L.Sy.p, L.Kyp, nkb, (const float *)L.E.p);
    SC_HIP(I, hipGetLastError());
    return SC_OK;
}

} // namespace sc
