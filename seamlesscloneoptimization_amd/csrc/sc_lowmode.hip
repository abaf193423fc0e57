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
// and the factor is below 1e-4 beyond the first ~n/64 modes per direction (K = lowmode_count(n) modes are kept; measured
// residual against all modes < 0.002 grey levels).  Such a field is very smooth -- its shortest
// wavelength is 128 pixels -- so both the projection and the expansion go through a coarse grid of NODES, every 8th field
// row and column, with bilinear interpolation in between (and its transpose for the projection):
//   k_lm_restrict : every 8 x 8 cell of U sends its hat-weighted sums to its four corner nodes        (one pass over U)
//   k_lm_cproject : V = node values;  parts of Uh[k][l] = sum_YX SyN[Y][k] V[Y][X] SxN[X][l], one per column tile and row split
//   k_lm_cexpand  : Chat = R (.) (the parts, added in order);  CN[Y][X] = sum_kl SyN[Y][k] Chat[k][l] SxN[X][l]: the correction at the nodes
//   post-process  : out = clamp/truncate( U[y][x] + bilinear(CN) )                                      (sc_kernels.hip)
// SyN[Y][k] = sin(pi 8Y (k+1)/(h+1)) are the rows of the DST matrix at the nodes (field row 8Y = interior index 8Y - 1).
// Representing a sine of mode k by linear interpolation from every 8th sample is off by at most 8 (pi (k+1)/(n+1))^2:
// 2e-5 for the first mode and 2e-2 for mode n/64, whose share of the correction is itself ~1e-3 -- measured: the residual
// against the full-table correction does not move in the 4th decimal (oracle/lowmode_np.py, hat = 1 vs 8).
// Tables: built on the device in double (sinpi with exact integer argument reduction); R = (den_exact / den_float - 1) *
// 4/((w+1)(h+1)) built on the host in double with the reference's float expressions.
// All sums run in a fixed order (no float atomics): results are reproducible bit for bit, alone or in a group.
#include "sc_instance.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace sc {

constexpr int LM_KB = 32;          // modes per register block
constexpr int LM_HAT = 8;          // node spacing (field rows / columns)
constexpr int LM_NPW = 8;          // node rows per wave in k_lm_cexpand
constexpr int LM_RS = 4;           // row splits of the coarse projection (workgroups per column tile and channel)

int lowmode_count(int n)
{
    int k = (n + 63) / 64;
    k = (std::max(k, 8) + 7) & ~7;
    return std::min(std::min(k, n), 256);
}

// S[r][l] = sin(pi (LM_HAT r)(l+1) / (n+1)), r < rows, l < K: the sine of mode l at the node in FIELD position LM_HAT r
// (interior index LM_HAT r - 1; node 0 is the Dirichlet ring, where every mode vanishes; the last node may lie past the
// far ring: the analytic continuation).  Row pitch Kp floats, pad columns zero.  Double, exact integer argument reduction.
__global__ __launch_bounds__(256) void k_lm_table(float *__restrict__ S, int n, int rows, int K, int Kp)
{
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)rows * Kp) return;
    const int r = (int)(id / Kp), l = (int)(id % Kp);
    float v = 0.f;
    if (l < K) {
        const long m = 2L * (n + 1);
        const long q = ((long)LM_HAT * r * (l + 1)) % m;          // sin(pi q / (n+1)) has period 2(n+1) in q
        v = (float)sinpi((double)q / (double)(n + 1));
    }
    S[id] = v;
}

// the tables of every member of a size class in one launch: blockIdx.y = 2 * member + direction (0: Sx, 1: Sy), into the member's
// own tables (RagMember::lm_Sx / lm_Sy, row pitch Kxp / Kyp = the class's)
__global__ __launch_bounds__(256) void k_lm_table_rag(const RagMember *__restrict__ rag, int Kxp, int Kyp)
{
    const RagMember &m = rag[blockIdx.y >> 1];
    const bool ydir = (blockIdx.y & 1) != 0;
    float *__restrict__ S = const_cast<float *>(ydir ? m.lm_Sy : m.lm_Sx);
    const int n = (ydir ? m.H : m.W) - 2, rows = ydir ? m.lm_ny : m.lm_nx, K = ydir ? m.lm_Ky : m.lm_Kx, Kp = ydir ? Kyp : Kxp;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < (long)rows * Kp; id += (long)gridDim.x * 256) {
        const int r = (int)(id / Kp), l = (int)(id % Kp);
        float v = 0.f;
        if (l < K) {
            const long mm = 2L * (n + 1);
            const long q = ((long)LM_HAT * r * (l + 1)) % mm;
            v = (float)sinpi((double)q / (double)(n + 1));
        }
        S[id] = v;
    }
}

// Restriction with the transpose of the bilinear interpolation: every 8 x 8 cell of the field (field rows 8Yc.., columns
// 8Xc..) sends its hat-weighted sums to its four corner nodes: Cell[c][Yc][Xc] = (a00, a01, a10, a11) for the nodes
// (Yc,Xc), (Yc,Xc+1), (Yc+1,Xc), (Yc+1,Xc+1).  Interior values only (ring and pads count as zero).  One pass over U:
// lane = 8 columns of one cell (two 16-byte loads per row, 8 rows in flight), no communication between lanes.
__global__ __launch_bounds__(256) void k_lm_restrict(Field U, float4 *__restrict__ Cell, int cells_x, int cells_y)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int Xc = blockIdx.x * 64 + lane, Yc = blockIdx.y * 4 + wv;
    if (Xc >= cells_x || Yc >= cells_y) return;
    const int x0 = 8 * Xc, y0 = 8 * Yc;
    const float *__restrict__ u = U.at(c) + x0;
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    float4 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = min(y0 + i, U.H - 1);                       // clamped address (x0 + 7 < pitch always), masked below
        lo[i] = *reinterpret_cast<const float4 *>(u + (size_t)y * U.pitch);
        hi[i] = *reinterpret_cast<const float4 *>(u + (size_t)y * U.pitch + 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = y0 + i;
        float v[8] = { lo[i].x, lo[i].y, lo[i].z, lo[i].w, hi[i].x, hi[i].y, hi[i].z, hi[i].w };
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = (x0 + j >= 1) && (x0 + j <= U.W - 2);
            const float val = in ? v[j] : 0.f;
            r0 = __builtin_fmaf(1.0f - 0.125f * j, val, r0);
            r1 = __builtin_fmaf(0.125f * j, val, r1);
        }
        if (y >= 1 && y <= U.H - 2) {
            a00 = __builtin_fmaf(1.0f - 0.125f * i, r0, a00); a01 = __builtin_fmaf(1.0f - 0.125f * i, r1, a01);
            a10 = __builtin_fmaf(0.125f * i, r0, a10);        a11 = __builtin_fmaf(0.125f * i, r1, a11);
        }
    }
    Cell[((size_t)c * cells_y + Yc) * cells_x + Xc] = make_float4(a00, a01, a10, a11);
}

// Coarse projection.  Node value V[Y][X] = the four cell shares that meet there; T[k][X] = sum_Y SyN[Y][k] V[Y][X];
// The same cell shares from the parts the final level-0 multigrid launch left (k_cycle0, `bands`): cell row Yc receives part A of
// the bands that start in it and part B of the bands that start in the cell row above; map[Yc][e] = 2 * band row + part, or -1,
// in ascending order (built on the host from the launch's tiling) -- a fixed order of at most four additions.
// rag (here and in the two kernels below): a size class -- strides (cells_x, cells_y, C, Kxp, Kyp, npitch) and grids are the class's,
// the extents, the tables and the split of the sums are member (channel / 3)'s own, so that every sum runs in the order of its solo run
__global__ __launch_bounds__(256) void k_lm_bands_to_cells(const float4 *__restrict__ bands, int band_rows, const int *__restrict__ map,
                                                           float4 *__restrict__ Cell, int cells_x, int cells_y, int W,
                                                           const RagMember *__restrict__ rag, int tiling)
{
    const int Xc = blockIdx.x * 256 + threadIdx.x, Yc = blockIdx.y, c = blockIdx.z;
    if (rag) {
        const RagMember &m = rag[c / 3];
        if (Yc >= m.lm_cells_y) return;
        map = m.lm_map[tiling]; W = m.W;
    }
    if (Xc >= cells_x) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = map[4 * Yc + e];
        if (m < 0 || LM_HAT * Xc >= W) continue;       // cells right of the field: pad columns, nothing was written for them
        const float4 v = bands[2 * (((size_t)c * band_rows + (m >> 1)) * cells_x + Xc) + (m & 1)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    Cell[((size_t)c * cells_y + Yc) * cells_x + Xc] = s;
}

// Upart[part][k][l] = sum_{X in tile} T[k][X] SxN[X][l] (part = column tile x row split; the expansion kernel adds the parts
// in a fixed order).  Workgroup = 64 node columns (lane = X), the four waves split the node rows and meet in LDS.
// (RAG a template parameter since late round 5: with the class's overrides in the one kernel the same-size launches ran 45 -> 63 us --
//  k_lm_cexpand 38 -> 64 -- although their `rag` was null: +0.5 % on the 2048^2 step, found by building round 4's tree beside this one)
// The sine tables are read-only for the whole launch and a wave reads one row at a time: through a constant-address-space pointer the
// loads are scalar (SGPRs) whatever the pointer's origin -- a kernel argument marked __restrict__ gets that by itself, a pointer loaded
// from a class's member table does not (the compiler must assume the kernel's own stores may alias it: vector loads, 189 VGPRs).
typedef const float __attribute__((address_space(4))) *lm_const_row;
template <bool RAG>
__global__ __launch_bounds__(256) void k_lm_cproject(const float4 *__restrict__ Cell, int cells_x, int cells_y, int nx, int ny, int C,
                                                      const float *__restrict__ SyN, int Kyp, const float *__restrict__ SxN, int Kxp,
                                                      float *__restrict__ Upart, const RagMember *__restrict__ rag)
{
    __shared__ float red[4][LM_KB][64];
    __shared__ float Tt[LM_KB][64 + 1];
    __shared__ float St[64][LM_KB + 1];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = threadIdx.x;
    // blockIdx.y = row split x block of LM_KB y-modes (round 4: the blocks used to be a loop inside the workgroup -- a tall ROI,
    // 1000 x 8000, has two column tiles and 8 blocks: 24 workgroups ran 188 us)
    const int nkb = (Kyp + LM_KB - 1) / LM_KB;
    const int xt = blockIdx.x, rs = blockIdx.y / nkb, c = blockIdx.z;
    int nxt = gridDim.x, nrs = gridDim.y / nkb;
    const float4 *__restrict__ cell = Cell + (size_t)c * cells_y * cells_x;      // (strides: before a member's own cell-row count replaces cells_y)
    if constexpr (RAG) {
        const RagMember &m = rag[c / 3];
        nxt = m.lm_nxt; nrs = m.lm_nrs;
        if (xt >= nxt || rs >= nrs) return;                     // block-uniform, before any barrier
        nx = m.lm_nx; ny = m.lm_ny; cells_y = m.lm_cells_y; SyN = m.lm_Sy; SxN = m.lm_Sx;
    }
    const int part = rs * nxt + xt;                              // this workgroup's partial product
    const int X = xt * 64 + lane;
    const int rows_per = (ny + 4 * nrs - 1) / (4 * nrs), Ya = (rs * 4 + wv) * rows_per, Yb = min(ny, Ya + rows_per);
    // clamped cell coordinates, masked values: the loads stay branch-free and all of a batch are in flight together
    const int Xc0 = min(X, cells_x - 1), Xc1 = min(max(X - 1, 0), cells_x - 1);
    const bool mx0 = X < nx && X < cells_x, mx1 = X < nx && X >= 1 && X - 1 < cells_x;
    {
        const int kb = (blockIdx.y % nkb) * LM_KB;
        float acc[LM_KB];
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) acc[k] = 0.f;
        for (int Yq = Ya; Yq < Yb; Yq += 8) {
            float v[8];
            float4 c00[8], c01[8], c10[8], c11[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int Y = Yq + q, Yc0 = min(Y, cells_y - 1), Yc1 = min(max(Y - 1, 0), cells_y - 1);
                c00[q] = cell[(size_t)Yc0 * cells_x + Xc0]; c01[q] = cell[(size_t)Yc0 * cells_x + Xc1];
                c10[q] = cell[(size_t)Yc1 * cells_x + Xc0]; c11[q] = cell[(size_t)Yc1 * cells_x + Xc1];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int Y = Yq + q;
                const bool my0 = Y < Yb && Y < cells_y, my1 = Y < Yb && Y >= 1 && Y - 1 < cells_y;
                const float s00 = (my0 && mx0) ? c00[q].x : 0.f, s01 = (my0 && mx1) ? c01[q].y : 0.f;
                const float s10 = (my1 && mx0) ? c10[q].z : 0.f, s11 = (my1 && mx1) ? c11[q].w : 0.f;
                v[q] = (s00 + s01) + (s10 + s11);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int Y = min(Yq + q, ny - 1);               // rows past Yb carry v = 0
                const lm_const_row s = (lm_const_row)(SyN + (size_t)Y * Kyp + kb);
#pragma unroll
                for (int k = 0; k < LM_KB; ++k) acc[k] = __builtin_fmaf(s[k], v[q], acc[k]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) red[wv][k][lane] = acc[k];
        __syncthreads();
        for (int i = t; i < LM_KB * 64; i += 256) {
            const int k = i >> 6, xx = i & 63;
            Tt[k][xx] = ((red[0][k][xx] + red[1][k][xx]) + (red[2][k][xx] + red[3][k][xx]));
        }
        __syncthreads();
        const int tk = t >> 4, tl = t & 15;                      // this thread's 2 x 2 block of the 32 x 32 product
        for (int lb = 0; lb < Kxp; lb += LM_KB) {
            __syncthreads();
            for (int i = t; i < 64 * LM_KB; i += 256) {
                const int xx = i >> 5, l = i & 31;
                St[xx][l] = SxN[(size_t)min(xt * 64 + xx, nx - 1) * Kxp + lb + l];          // columns past nx carry T = 0
            }
            __syncthreads();
            float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 8
            for (int xx = 0; xx < 64; ++xx) {
                const float s0 = St[xx][2 * tl], s1 = St[xx][2 * tl + 1];
                const float t0 = Tt[2 * tk][xx], t1 = Tt[2 * tk + 1][xx];
                a00 = __builtin_fmaf(t0, s0, a00); a01 = __builtin_fmaf(t0, s1, a01);
                a10 = __builtin_fmaf(t1, s0, a10); a11 = __builtin_fmaf(t1, s1, a11);
            }
            float *__restrict__ o = Upart + (((size_t)part * C + c) * Kyp + kb + 2 * tk) * Kxp + lb + 2 * tl;
            o[0] = a00; o[1] = a01; o[Kxp] = a10; o[Kxp + 1] = a11;
        }
    }
}

// The projection's parts added up once, in order, into part 0 (round 4).  k_lm_cexpand adds them itself -- every one of its
// workgroups all of them: fine for the 16 parts of a 2048^2 ROI, 67 us for the 64 parts x 32 x 256 modes of an 8000 x 1000 one.
__global__ __launch_bounds__(256) void k_lm_parts_sum(float *__restrict__ Upart, int nparts, size_t per_part, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 4
    for (int q = 0; q < nparts; ++q) s += Upart[(size_t)q * per_part + i];
    Upart[i] = s;
}

// Coarse expansion: Chat[k][l] = R[k][l] * (sum of the projection's parts, in order);  E[k][X] = sum_l Chat[k][l] SxN[X][l];
// CN[c][Y][X] = sum_k SyN[Y][k] E[k][X] -- the correction at the nodes.  Workgroup = 64 node columns x (4 waves x LM_NPW node rows); E of the 64 columns is formed once per workgroup
// (wave v takes 8 of every 32 modes, through LDS); then lane = X with the E column in registers, SyN rows as scalar loads.
template <bool RAG>
__global__ __launch_bounds__(256) void k_lm_cexpand(const float *__restrict__ Upart, int nparts, int C, const float *__restrict__ R,
                                                    const float *__restrict__ SxN, int Kxp,
                                                    const float *__restrict__ SyN, int Kyp, int nx, int ny, int npitch,
                                                    float *__restrict__ CN, const RagMember *__restrict__ rag)
{
    __shared__ __attribute__((aligned(16))) float Ch[LM_KB][LM_KB + 4];
    __shared__ float Es[LM_KB][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = threadIdx.x;
    const int c = blockIdx.z;
    float *__restrict__ const cn_plane = CN + (size_t)c * ny * npitch;      // (strides: the launch's ny, before a member's own replaces it)
    if constexpr (RAG) {
        const RagMember &m = rag[c / 3];
        if ((int)blockIdx.x >= m.lm_nxt || (int)blockIdx.y * 4 * LM_NPW >= m.lm_ny) return;      // block-uniform, before any barrier
        nparts = m.lm_nparts; R = m.lm_R; SxN = m.lm_Sx; SyN = m.lm_Sy; nx = m.lm_nx; ny = m.lm_ny;
    }
    const int X = blockIdx.x * 64 + lane, Xs = min(X, nx - 1);
    const int Y0 = (blockIdx.y * 4 + wv) * LM_NPW;
    float cn[LM_NPW];
#pragma unroll
    for (int q = 0; q < LM_NPW; ++q) cn[q] = 0.f;
    for (int kb = 0; kb < Kyp; kb += LM_KB) {
        float ek[LM_KB / 4];                                     // this wave's 8 modes of the block
#pragma unroll
        for (int q = 0; q < LM_KB / 4; ++q) ek[q] = 0.f;
        for (int lb = 0; lb < Kxp; lb += LM_KB) {
            __syncthreads();
            {
                float sum[LM_KB * LM_KB / 256];
#pragma unroll
                for (int j = 0; j < LM_KB * LM_KB / 256; ++j) sum[j] = 0.f;
#pragma unroll 2
                for (int q = 0; q < nparts; ++q) {
                    const float *__restrict__ up = Upart + (((size_t)q * C + c) * Kyp + kb) * Kxp + lb;
#pragma unroll
                    for (int j = 0; j < LM_KB * LM_KB / 256; ++j) {
                        const int idx = t + 256 * j;
                        sum[j] += up[(size_t)(idx >> 5) * Kxp + (idx & 31)];
                    }
                }
#pragma unroll
                for (int j = 0; j < LM_KB * LM_KB / 256; ++j) {
                    const int idx = t + 256 * j;
                    Ch[idx >> 5][idx & 31] = sum[j] * R[(size_t)(kb + (idx >> 5)) * Kxp + lb + (idx & 31)];
                }
            }
            float sx[LM_KB];
#pragma unroll
            for (int q = 0; q < LM_KB / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(SxN + (size_t)Xs * Kxp + lb + 4 * q);
                sx[4 * q] = v.x; sx[4 * q + 1] = v.y; sx[4 * q + 2] = v.z; sx[4 * q + 3] = v.w;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < LM_KB / 4; ++q) {
                float a = ek[q];
#pragma unroll
                for (int l = 0; l < LM_KB; ++l) a = __builtin_fmaf(Ch[wv * (LM_KB / 4) + q][l], sx[l], a);
                ek[q] = a;
            }
        }
#pragma unroll
        for (int q = 0; q < LM_KB / 4; ++q) Es[wv * (LM_KB / 4) + q][lane] = ek[q];
        __syncthreads();
        float e[LM_KB];
#pragma unroll
        for (int k = 0; k < LM_KB; ++k) e[k] = Es[k][lane];
#pragma unroll
        for (int q = 0; q < LM_NPW; ++q) {
            const int Y = min(Y0 + q, ny - 1);
            const lm_const_row s = (lm_const_row)(SyN + (size_t)Y * Kyp + kb);
            float a = cn[q];
#pragma unroll
            for (int k = 0; k < LM_KB; ++k) a = __builtin_fmaf(s[k], e[k], a);
            cn[q] = a;
        }
    }
    if (X >= npitch) return;
    float *__restrict__ o = cn_plane + X;
#pragma unroll
    for (int q = 0; q < LM_NPW; ++q)
        if (Y0 + q < ny) o[(size_t)(Y0 + q) * npitch] = (X < nx) ? cn[q] : 0.f;
}

// Out = U + bilinearly interpolated node correction on the interior, as float (diagnostic hook; the clone itself applies
// the correction inside the post-process, sc_kernels.hip, with the same formula: lm_bilinear)
__global__ __launch_bounds__(256) void k_lm_apply(Field U, Field Out, LmNodes lm)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    if (x < 1 || x > U.W - 2 || y < 1 || y > U.H - 2) return;
    const size_t o = (size_t)y * U.pitch + x;
    Out.at(c)[o] = U.at(c)[o] + lm_bilinear(lm, c, x, y);
}

// ---------------------------------------------------------------------------------------------------------- host side

// R[k][l] = (den_exact / den_float - 1) * 4 / ((w+1)(h+1)).  den_float follows the reference to the letter:
// filter_X[i] = (float)(2.0 * cos(PI/(n+1.0) * (i+1.0))) with the FLOAT literal PI (seamlessClone_imp.h:17, .cpp:596-599),
// den = filter_X[i] + filter_Y[j] - 4 in float (:1651-1653).  den_exact = -4 (sin^2(a/2) + sin^2(b/2)) in double.
// Returns false when the reference's arithmetic is SINGULAR for this size: beyond ~12 870 pixels in both directions
// 2 cos(pi/(n+1)) rounds to 2.0f, the float denominator of the lowest mode is zero and the reference divides by zero
// (its result is NaN).  There is nothing to reproduce then: the caller applies no correction (exact system).
static bool build_ratio(int w, int h, int Kx, int Ky, int Kxp, float *R, double &max_ratio);
bool lowmode_ratio(int w, int h, int Kx, int Ky, int Kxp, float *R, double &max_ratio) { return build_ratio(w, h, Kx, Ky, Kxp, R, max_ratio); }

// row splits of the coarse projection: as few as fill the chip (every split is one more part for k_lm_cexpand to add: a wide ROI, 8000 x
// 1000, has 16 column tiles and needs none), at most LM_RS (what the parts buffer holds) -- from the geometry alone, as for one clone: a
// group member's bytes are the clone's own
int lowmode_projection_splits(int nxt, int nkb)
{
    int nrs = 1;
    while (nrs < LM_RS && nxt * nrs * nkb * 3 < 192) nrs *= 2;
    return nrs;
}

void launch_lm_tables_rag(const RagMember *rag, int members, int max_rows, int Kxp, int Kyp, hipStream_t s)
{
    const unsigned gx = (unsigned)std::max<size_t>(1, ((size_t)max_rows * (size_t)std::max(Kxp, Kyp) + 255) / 256);
    hipLaunchKernelGGL(k_lm_table_rag, dim3(gx, 2 * members), dim3(256), 0, s, rag, Kxp, Kyp);
}

static bool build_ratio(int w, int h, int Kx, int Ky, int Kxp, float *R, double &max_ratio)
{
    max_ratio = 0.0;
    const double PIf = (double)3.14159265358979323846f;
    float fx[256], fy[256];                         // (lowmode_count: at most 256 modes per direction)
    double sa2[256], sb2[256];                      // sin^2 of the half angles: one sine per mode and direction, not per table entry
    if (Kx > 256 || Ky > 256) return false;
    for (int i = 0; i < Kx; ++i) {
        fx[i] = (float)(2.0 * std::cos(PIf / (w + 1.0) * (i + 1.0)));
        const double sa = std::sin(0.5 * M_PI * (i + 1.0) / (w + 1.0));
        sa2[i] = sa * sa;
    }
    for (int j = 0; j < Ky; ++j) {
        fy[j] = (float)(2.0 * std::cos(PIf / (h + 1.0) * (j + 1.0)));
        const double sb = std::sin(0.5 * M_PI * (j + 1.0) / (h + 1.0));
        sb2[j] = sb * sb;
    }
    const double scale = 4.0 / ((w + 1.0) * (h + 1.0));
    if (R) {
        // the table: every entry, branch-free so that the compiler vectorises the divisions (a new ROI size pays this once, inside the
        // first batch call that meets it: 12 us at 2100 x 2100 before, ~4 now)
        for (int j = 0; j < Ky; ++j) {
            float *const Rj = R + (size_t)j * Kxp;
            const float fyj = fy[j];
            const double sbj = sb2[j];
            double mr = 0.0;
            for (int i = 0; i < Kx; ++i) {
                const double den_e = -4.0 * (sa2[i] + sbj);
                const float den_f = (fx[i] + fyj) - 4.0f;
                const bool ok = den_f < 0.0f;
                const double q = den_e / (ok ? (double)den_f : -1.0) - 1.0;
                Rj[i] = ok ? (float)(q * scale) : 0.0f;
                const double aq = std::fabs(q);
                mr = (ok && aq > mr) ? aq : mr;
            }
            for (int i = Kx; i < Kxp; ++i) Rj[i] = 0.f;
            max_ratio = std::max(max_ratio, mr);
        }
    } else {
        // the statistic alone (plan_size): |den_e / den_f - 1| = |den_e - den_f| / |den_f|, and the float denominator is off by a few
        // units in its last place at most -- 2 cos rounds to float (6e-8 each), the float sum of two values near 2 (1.2e-7), the float
        // literal PI (relative 5.6e-8 of den) -- so an entry with |den_e| (1 - 1e-6) >= E / max so far cannot raise the maximum; den_e
        // grows along a row and down a column: the search visits a handful of entries instead of all Kx Ky
        const double E = 4e-7;
        for (int j = 0; j < Ky; ++j) {
            if (max_ratio > 0.0 && 4.0 * sb2[j] * (1.0 - 1e-6) * max_ratio >= E) break;       // this row's FIRST entry is already out
            for (int i = 0; i < Kx; ++i) {
                const double den_e = -4.0 * (sa2[i] + sb2[j]);
                if (max_ratio > 0.0 && -den_e * (1.0 - 1e-6) * max_ratio >= E) break;
                const float den_f = (fx[i] + fy[j]) - 4.0f;
                if (den_f < 0.0f) max_ratio = std::max(max_ratio, std::fabs(den_e / (double)den_f - 1.0));
            }
        }
    }
    return (fx[0] + fy[0]) - 4.0f < 0.0f;          // the lowest mode has the denominator closest to zero
}

// (Re)builds the tables for the fields currently bound to the instance; no-op when the geometry is unchanged.
// A size class: the members' tables are on the device already (rag_begin_table / rag_begin_builds, sc_ragged.cpp); what is left are the class-sized work
// buffers.  L's geometry holds the class's strides and maxima.
static int lm_prepare_rag(Instance *I)
{
    LowMode &L = I->lm;
    const RagState &R = I->rag;
    const int H = I->F.H, C = I->F.C, pitch = I->F.pitch;
    const int cells_x = pitch / LM_HAT, cells_y = (H + LM_HAT - 1) / LM_HAT;
    const int npitch = round_up(R.max_nx, 4);
    int rc;
    if ((rc = ensure(I, L.P, sizeof(float4) * (size_t)C * cells_y * cells_x))) return rc;
    if ((rc = ensure(I, L.E, sizeof(float) * (size_t)R.max_nxt * LM_RS * C * R.Kyp * R.Kxp))) return rc;
    if ((rc = ensure(I, L.CN, sizeof(float) * (size_t)C * R.max_ny * npitch))) return rc;
    L.C = C;
    L.w = L.h = 0;                             // (the views Sx / Sy / R are not used by a class: the next ordinary clone binds its own)
    L.singular = false; L.max_ratio = R.max_ratio;
    L.Kx = L.Ky = 0; L.Kxp = R.Kxp; L.Kyp = R.Kyp; L.nx = R.max_nx; L.ny = R.max_ny; L.npitch = npitch;
    return SC_OK;
}

static int lm_prepare(Instance *I)
{
    if (I->rag.dev) return lm_prepare_rag(I);
    LowMode &L = I->lm;
    const int W = I->F.W, H = I->F.H, C = I->F.C, pitch = I->F.pitch;
    const int w = W - 2, h = H - 2;
    const int Kx = lowmode_count(w), Ky = lowmode_count(h);
    const int Kxp = round_up(Kx, LM_KB), Kyp = round_up(Ky, LM_KB);
    const int cells_x = pitch / LM_HAT, cells_y = (H + LM_HAT - 1) / LM_HAT;      // 8 x 8 cells covering the field
    const int nx = ((W - 2) >> 3) + 2, ny = ((H - 2) >> 3) + 2;                   // nodes: every cell with interior points has both its corners
    const int npitch = round_up(nx, 4);
    const int nxt = (nx + 63) / 64;
    int rc;
    if ((rc = ensure(I, L.P, sizeof(float4) * (size_t)C * cells_y * cells_x))) return rc;           // cell shares
    if ((rc = ensure(I, L.E, sizeof(float) * (size_t)nxt * LM_RS * C * Kyp * Kxp))) return rc;       // partial products of the coarse projection
    if ((rc = ensure(I, L.CN, sizeof(float) * (size_t)C * ny * npitch))) return rc;
    L.C = C;
    if (L.w == w && L.h == h && L.Sx.p && L.Sy.p && L.R.p) return SC_OK;
    L.w = 0;
    LowMode::Tables *T = nullptr, *victim = nullptr;
    for (LowMode::Tables &t : L.tables) {
        if (t.R.p && t.w == w && t.h == h) { T = &t; break; }
        if (!victim || t.used < victim->used) victim = &t;
    }
    if (!T) {                                  // a size not in the cache: build into the least recently used entry
        T = victim;
        I->info.new_size = 1;
        if (T->ev) SC_HIP(I, hipEventSynchronize(T->ev));       // its staging fed an upload once: long complete
        else SC_HIP(I, hipEventCreateWithFlags(&T->ev, hipEventDisableTiming));
        T->w = 0;
        if ((rc = ensure(I, T->Sx, sizeof(float) * (size_t)nx * Kxp))) return rc;
        if ((rc = ensure(I, T->Sy, sizeof(float) * (size_t)ny * Kyp))) return rc;
        if ((rc = ensure(I, T->R, sizeof(float) * (size_t)Kyp * Kxp))) return rc;
        if ((rc = ensure_pinned(I, T->hR, sizeof(float) * (size_t)Kyp * Kxp))) return rc;
        std::memset(T->hR.p, 0, sizeof(float) * (size_t)Kyp * Kxp);
        T->singular = !build_ratio(w, h, Kx, Ky, Kxp, (float *)T->hR.p, T->max_ratio);
        SC_HIP(I, hipMemcpyAsync(T->R.p, T->hR.p, sizeof(float) * (size_t)Kyp * Kxp, hipMemcpyHostToDevice, I->stream));
        SC_HIP(I, hipEventRecord(T->ev, I->stream));
        hipLaunchKernelGGL(k_lm_table, dim3((unsigned)(((size_t)nx * Kxp + 255) / 256)), dim3(256), 0, I->stream, (float *)T->Sx.p, w, nx, Kx, Kxp);
        hipLaunchKernelGGL(k_lm_table, dim3((unsigned)(((size_t)ny * Kyp + 255) / 256)), dim3(256), 0, I->stream, (float *)T->Sy.p, h, ny, Ky, Kyp);
        SC_HIP(I, hipGetLastError());
        T->w = w; T->h = h;
    }
    T->used = ++L.tick;
    L.Sx = T->Sx; L.Sy = T->Sy; L.R = T->R;      // views
    L.singular = T->singular; L.max_ratio = T->max_ratio;
    L.w = w; L.h = h; L.Kx = Kx; L.Ky = Ky; L.Kxp = Kxp; L.Kyp = Kyp; L.nx = nx; L.ny = ny; L.npitch = npitch;
    return SC_OK;
}

// Which parts make up each cell row.  A level-0 launch with `sweeps` sweeps leaves, per 8-row band b (tile row b / 8, wave b % 8)
// and 8-column cell, part 2b for the cell row the band starts in and part 2b + 1 for the next one (k_cycle0, `bands`);
// m[4 Yc + e] lists the parts of cell row Yc in ascending order, -1 = none.  False: a tiling with more than four parts in a
// cell row (none of the instantiated ones).  Pure host arithmetic on the launch geometry (cycle0_row_geometry).
bool lowmode_part_map(int H, int sweeps, std::vector<int> &m, int &band_rows)
{
    m.assign(4 * (size_t)((H + LM_HAT - 1) / LM_HAT), -1);
    return lowmode_part_map(H, sweeps, m.data(), band_rows);
}

// (m: 4 ints per cell row, every one -1 on entry; parts arrive in ascending order, a cell row's list is full when its last slot is taken)
bool lowmode_part_map(int H, int sweeps, int *m, int &band_rows)
{
    const int cells_y = (H + LM_HAT - 1) / LM_HAT;
    int nby, step, hy;
    cycle0_row_geometry(H, sweeps, nby, step, hy);
    band_rows = nby * 8;
    bool fits = true;
    auto add = [&](int Yc, int v) {
        if (Yc < 0 || Yc >= cells_y) return;
        int *const e = m + 4 * (size_t)Yc;
        const int k = e[0] < 0 ? 0 : e[1] < 0 ? 1 : e[2] < 0 ? 2 : e[3] < 0 ? 3 : 4;
        if (k < 4) e[k] = v; else fits = false;
    };
    for (int b = 0; b < band_rows; ++b) {                       // ascending band rows: a fixed order of additions per cell
        const int y0 = (b >> 3) * step - hy + 8 * (b & 7);
        const int Yc0 = y0 >= 0 ? y0 >> 3 : -((-y0 + 7) >> 3);  // floor(y0 / 8)
        add(Yc0, 2 * b); add(Yc0 + 1, 2 * b + 1);
    }
    return fits;
}

// Host check of the map against the launch geometry it describes (sc_hip_selftest_host): every interior field row lies in the
// exact output rows of exactly one band, and the part that band forms for the row's cell row is in the map.  0 = fine.
int lowmode_part_map_selftest()
{
    const int sizes[] = { 3, 4, 9, 10, 11, 53, 54, 55, 64, 105, 300, 517, 1000, 1025, 2048, 2049, 4096, 6000 };
    for (int sweeps : { 1, 2, 3, 4 })
        for (int H : sizes) {
            std::vector<int> m;
            int band_rows = 0;
            if (!lowmode_part_map(H, sweeps, m, band_rows)) return 1;
            int nby, step, hy;
            cycle0_row_geometry(H, sweeps, nby, step, hy);
            for (int y = 1; y <= H - 2; ++y) {
                int owners = 0, part = -1;
                for (int b = 0; b < band_rows; ++b) {
                    const int y0 = (b >> 3) * step - hy + 8 * (b & 7), yr = y - ((b >> 3) * step - hy);      // row inside the tile
                    if (y < y0 || y >= y0 + 8 || yr < hy || yr >= 64 - hy) continue;
                    ++owners;
                    const int Yc0 = y0 >= 0 ? y0 >> 3 : -((-y0 + 7) >> 3);
                    part = 2 * b + ((y >> 3) == Yc0 ? 0 : 1);
                }
                if (owners != 1) return 2;
                bool listed = false;
                for (int e = 0; e < 4; ++e) listed = listed || m[4 * (y >> 3) + e] == part;
                if (!listed) return 3;
            }
            for (size_t i = 0; i + 1 < m.size(); ++i)               // ascending inside a cell row, -1 only at the end
                if ((i & 3) != 3 && m[i + 1] != -1 && (m[i] == -1 || m[i] >= m[i + 1])) return 4;
        }
    return 0;
}

// Buffer for the cell-share parts of a final level-0 launch with `sweeps` sweeps on the instance's current fields (k_cycle0's
// `bands` argument), or nullptr when no correction will be applied.  The caller passes it to that launch and then calls
// lowmode_bands_written(); the next lowmode_nodes() builds the cells from the parts instead of reading the field again.
float4 *lowmode_bands_buffer(Instance *I, int sweeps)
{
    LowMode &L = I->lm;
    L.bands_of = nullptr;
    if (!wants_float_tables(I) || lm_prepare(I) != SC_OK || L.singular) return nullptr;
    const int H = I->F.H, C = I->F.C, cells_x = I->F.pitch / LM_HAT, cells_y = (H + LM_HAT - 1) / LM_HAT;
    int nby, step, hy;
    cycle0_row_geometry(H, sweeps, nby, step, hy);
    int band_rows = nby * 8;
    if (ensure(I, L.B, sizeof(float4) * 2 * (size_t)C * band_rows * cells_x) != SC_OK) return nullptr;
    if (I->rag.dev) {                                       // a size class: every member's two maps are in its table entry
        if (sweeps != 2 && sweeps != 4) return nullptr;
        L.rag_tiling = sweeps > 2;
        L.band_rows = band_rows;
        return (float4 *)L.B.p;
    }
    LowMode::PartMap &M = L.maps[sweeps > 2];              // the two tilings a solve uses (2 and 4 sweeps) keep a map each
    L.map_used = &M;
    if (M.H != H || M.sweeps != sweeps || !M.d.p) {
        M.H = 0;
        if (ensure(I, M.d, sizeof(int) * 4 * (size_t)cells_y) != SC_OK) return nullptr;
        if (ensure_pinned(I, M.h, sizeof(int) * 4 * (size_t)cells_y) != SC_OK) return nullptr;
        std::vector<int> m;
        if (!lowmode_part_map(H, sweeps, m, band_rows)) return nullptr;      // more than four parts per cell row: the separate pass serves
        std::memcpy(M.h.p, m.data(), sizeof(int) * m.size());
        if (hipMemcpyAsync(M.d.p, M.h.p, sizeof(int) * 4 * (size_t)cells_y, hipMemcpyHostToDevice, I->stream) != hipSuccess) return nullptr;
        M.H = H; M.sweeps = sweeps;
    }
    L.band_rows = band_rows;
    return (float4 *)L.B.p;
}

void lowmode_bands_written(Instance *I, const float *field) { I->lm.bands_of = field; }

// Can the correction of the NEXT cycle's result be taken from the current iterate?  The correction is linear in the field:
// corr(u_next) - corr(u_now) = corr(u_next - u_now), at most max_ratio x the low-mode content of the next cycle's update m.
// The stop rule accepts that cycle when m rho / (1 - rho) <= 0.1 update_tol with rho floored at 0.02, i.e. for m up to
// 4.9 update_tol (1.2 grey levels at the default 0.25; measured third-cycle updates are 0.06-0.09).  Worst case of the
// difference therefore: max_ratio x 4.9 x update_tol = 0.015 grey levels at 2048^2 (max_ratio 0.012), 0.036 at 4096^2
// (0.029); typical 0.001-0.003.  0: no correction is applied at all (exact tables, singular float tables, no unknowns);
// 1: yes, that worst case stays below 0.05 grey levels; 3: only if the judged cycle's MEASURED update m keeps max_ratio x m below
// the same 0.049 (the caller checks it beside the stop rule and repeats the cycle in the field-keeping form otherwise): ROIs
// whose float tables are off by more than 4 % in their lowest modes -- everything beyond ~5000^2, and single sizes from ~3000^2
// on where 2 cos(pi / (n + 1)) rounds unluckily (3120^2, 3330^2, 3470^2, 3610^2: those took the field-keeping path with its
// serial correction and separate post-process until round 4, +8-10 % per clone; measured third-cycle updates are 0.06-0.09,
// i.e. a difference of 0.004); 2: (lm_prepare failed) the field-keeping path.
int lowmode_early_kind(Instance *I, float update_tol)
{
    if (!wants_float_tables(I)) return 0;
    if (lm_prepare(I) != SC_OK) return 2;
    if (I->lm.singular) return 0;
    return I->lm.max_ratio * 4.9 * (double)update_tol <= 0.049 ? 1 : 3;
}

// Node corrections of the field U (the instance's current shape): what the post-process adds (lm.CN == nullptr: nothing).
int lowmode_nodes(Instance *I, const Field &U, LmNodes &lm, hipStream_t on)
{
    lm = LmNodes();
    hipStream_t const st = on ? on : I->stream;
    if (U.W < 3 || U.H < 3) return SC_OK;
    int rc = lm_prepare(I);
    if (rc) return rc;
    LowMode &L = I->lm;
    if (L.singular) return SC_OK;                  // the reference's float tables divide by zero at this size: exact system
    const int cells_x = U.pitch / LM_HAT, cells_y = (U.H + LM_HAT - 1) / LM_HAT, nxt = (L.nx + 63) / 64;
    float *upart = (float *)L.E.p;
    if (I->rag.dev) {
        // a size class: the same three launches on the class's grid; extents, tables and the split of every sum are each member's own
        const RagState &R = I->rag;
        if (!(L.bands_of && L.bands_of == U.p)) { I->err = "size class: the correction needs the cell shares of the level-0 launch"; return SC_ERR_BAD_ARG; }
        const int nkb = (R.Kyp + LM_KB - 1) / LM_KB;
        hipLaunchKernelGGL(k_lm_bands_to_cells, dim3((cells_x + 255) / 256, cells_y, U.C), dim3(256), 0, st, (const float4 *)L.B.p,
                           L.band_rows, (const int *)nullptr, (float4 *)L.P.p, cells_x, cells_y, U.W, R.dev, L.rag_tiling);
        L.bands_of = nullptr;
        hipLaunchKernelGGL(k_lm_cproject<true>, dim3(R.max_nxt, R.max_nrs * nkb, U.C), dim3(256), 0, st, (const float4 *)L.P.p, cells_x, cells_y, L.nx, L.ny, U.C,
                           (const float *)nullptr, R.Kyp, (const float *)nullptr, R.Kxp, upart, R.dev);
        hipLaunchKernelGGL(k_lm_cexpand<true>, dim3(R.max_nxt, (R.max_ny + 4 * LM_NPW - 1) / (4 * LM_NPW), U.C), dim3(256), 0, st, (const float *)upart,
                           1, U.C, (const float *)nullptr, (const float *)nullptr, R.Kxp, (const float *)nullptr, R.Kyp, L.nx, L.ny, L.npitch, (float *)L.CN.p, R.dev);
        SC_HIP(I, hipGetLastError());
        lm.CN = (const float *)L.CN.p;
        lm.ny = L.ny;
        lm.npitch = L.npitch;
        return SC_OK;
    }
    if (L.bands_of && L.bands_of == U.p) {       // the final level-0 launch left the cell shares in parts: no further pass over U
        hipLaunchKernelGGL(k_lm_bands_to_cells, dim3((cells_x + 255) / 256, cells_y, U.C), dim3(256), 0, st, (const float4 *)L.B.p,
                           L.band_rows, (const int *)L.map_used->d.p, (float4 *)L.P.p, cells_x, cells_y, U.W, (const RagMember *)nullptr, 0);
        L.bands_of = nullptr;  // used once: whoever touches the field afterwards need not know about the parts
    } else
    hipLaunchKernelGGL(k_lm_restrict, dim3((cells_x + 63) / 64, (cells_y + 3) / 4, U.C), dim3(256), 0, st, U, (float4 *)L.P.p, cells_x, cells_y);
    // row splits of the projection: as few as fill the chip (every split is one more part for k_lm_cexpand to add: a wide ROI, 8000 x
    // 1000, has 16 column tiles and needs none), at most LM_RS (what the parts buffer holds)
    const int nkb = (L.Kyp + LM_KB - 1) / LM_KB;
    const int nrs = lowmode_projection_splits(nxt, nkb);
    hipLaunchKernelGGL(k_lm_cproject<false>, dim3(nxt, nrs * nkb, U.C), dim3(256), 0, st, (const float4 *)L.P.p, cells_x, cells_y, L.nx, L.ny, U.C,
                       (const float *)L.Sy.p, L.Kyp, (const float *)L.Sx.p, L.Kxp, upart, (const RagMember *)nullptr);
    int nparts = nxt * nrs;
    if (nparts >= 32) {       // many parts (wide ROIs): one launch adds them, in the same order, instead of every expansion workgroup
        const size_t per_part = (size_t)U.C * L.Kyp * L.Kxp;
        hipLaunchKernelGGL(k_lm_parts_sum, dim3((unsigned)((per_part + 255) / 256)), dim3(256), 0, st, upart, nparts, per_part, per_part);
        nparts = 1;
    }
    hipLaunchKernelGGL(k_lm_cexpand<false>, dim3(nxt, (L.ny + 4 * LM_NPW - 1) / (4 * LM_NPW), U.C), dim3(256), 0, st, (const float *)upart,
                       nparts, U.C, (const float *)L.R.p, (const float *)L.Sx.p, L.Kxp, (const float *)L.Sy.p, L.Kyp, L.nx, L.ny, L.npitch, (float *)L.CN.p, (const RagMember *)nullptr);
    SC_HIP(I, hipGetLastError());
    lm.CN = (const float *)L.CN.p;
    lm.ny = L.ny;
    lm.npitch = L.npitch;
    return SC_OK;
}

// Out = U + correction as a float field (interior; ring and pads of Out are left as they are).  Out may not alias U.
int lowmode_correct(Instance *I, const Field &U, const Field &Out)
{
    LmNodes lm;
    int rc = lowmode_nodes(I, U, lm);
    if (rc || !lm.CN) return rc;
    hipLaunchKernelGGL(k_lm_apply, dim3((U.W + 255) / 256, U.H, U.C), dim3(256), 0, I->stream, U, Out, lm);
    field_moved(I);
    SC_HIP(I, hipGetLastError());
    return SC_OK;
}

} // namespace sc
