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
#include <stdlib.h>
#include "sc_wave.h"
#include "sc_mg_device.h"

namespace sc {

constexpr int TB_HX = 4; // column halo (one float4)

// R rows of one lane's float4 column, all requested at once.  Addresses are clamped into the plane
// rather than tested (a test around each load serialises the R fetches behind R waits); a clamped
// lane / row holds a value that is never used, because the ring is fixed and every update is masked
// to the interior, so out-of-range values only flow into other out-of-range values.
template <int R>
__device__ __forceinline__ void tb_load(const float *__restrict__ p, int P, int H, int x, int y0, float4 (&v)[R])
{
    const int xc = min(max(x, 0), P - 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);
        v[r] = *reinterpret_cast<const float4 *>(p + (size_t)yc * P + xc);
    }
}

// ---------------------------------------------------------------------------- red-black
// GEN = general-coefficient variant for the multigrid coarse levels (MGDim: modified stencil at
// the last column / row, division by the true diagonal); GEN = false is the exact level-0 form.
// FLAGS (multigrid fusions): TB_PROLONG adds the interpolated coarse correction P*E to U while
// loading it (replaces k_prolong_add), TB_MAXC also reduces max|P*E| per block into `partial`,
// TB_ZEROIN treats Uin as all-zero without reading it (first smoothing of a coarse correction).
constexpr int TB_PROLONG = 1, TB_MAXC = 2, TB_ZEROIN = 4;
constexpr int TB_TAG = 8;   // no effect on the code: a second symbol for the isolated roofline launches (see k_jacobi)
constexpr int TB_RAG = 16;  // the launch serves a size class (RagMember, sc_common.h): level `lev` of every member's own hierarchy

template <int T, int NW, int R, bool SOR, bool GEN, int FLAGS, int HXQ = 1>
__global__ __launch_bounds__(NW * 64) void k_rb_tb(Field Uin, Field Uout, Field F, float omega, MGGeom g, Field E,
                                                   float *__restrict__ partial, const RagMember *__restrict__ rag, int lev)
{
    constexpr int HY = 2 * T, RH = NW * R, HX = 4 * HXQ;   // HXQ halo lanes per side: 4 columns each
    static_assert(2 * T <= HX, "column halo too small for this depth");
    __shared__ float4 edge[2][NW][2][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR: row tests become scalar
    int W = Uin.W, H = Uin.H;
    const int P = Uin.pitch;
    // 1-D launch; neighbouring tiles are given to the same XCD so that they share its L2 (sc_wave.h)
    const int nbx = (W + (256 - 2 * HX) - 1) / (256 - 2 * HX), nby = (H + (RH - 2 * HY) - 1) / (RH - 2 * HY);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int bx = tile % nbx, by = (tile / nbx) % nby, c = tile / (nbx * nby);
    if constexpr ((FLAGS & TB_RAG) != 0) {      // the member's own level inside the class's strides; tiles beyond it leave (block-uniform, before any barrier)
        const RagMember &m = rag[c / 3];
        W = m.lw[lev]; H = m.lh[lev];
        if (bx * (256 - 2 * HX) >= W || by * (RH - 2 * HY) >= H) {
            if ((FLAGS & TB_MAXC) && threadIdx.x == 0) partial[tile] = 0.f;
            return;
        }
        g = m.g[lev];
        if (FLAGS & TB_PROLONG) E.H = m.lh[lev + 1];
    }
    const int x = bx * (256 - 2 * HX) - HX + 4 * lane;
    const int ry = by * (RH - 2 * HY) - HY;
    const int y0 = ry + wv * R;
    float4 u[R], f[R];
    if (FLAGS & TB_ZEROIN) {
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        tb_load<R>(Uin.at(c), P, H, x, y0, u);
    }
    tb_load<R>(F.at(c), P, H, x, y0, f);
    // coarse values the lane interpolates from (sc_mg_device.h), requested together with U and F
    constexpr bool PROL = (FLAGS & TB_PROLONG) != 0;
    static_assert(!PROL || R % 2 == 0, "the prolongation pairs fine rows");
    ProlongWindow<R, false, GEN> pw;       // a coarse level can have two tail points: LEFT = GEN; never composed here
    const ComposeArgs comp{};              // (composing level 3 into level 2's launch was measured neutral, sc_multigrid.cpp)
    if (PROL) prolong_load(pw, E, comp, c, x, y0);
    const bool x0ok = (x + 0 >= 1) && (x + 0 <= W - 2), x1ok = (x + 1 >= 1) && (x + 1 <= W - 2);
    const bool x2ok = (x + 2 >= 1) && (x + 2 <= W - 2), x3ok = (x + 3 >= 1) && (x + 3 <= W - 2);
    if (FLAGS & TB_PROLONG) {
        // u += P*E (one branch-free path for every lane, sc_mg_device.h)
        float m = prolong_apply(pw, g, comp, x, y0, W, H, u);
        if (FLAGS & TB_MAXC) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            __shared__ float red[NW];
            if (lane == 0) red[wv] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                float mm = red[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red[w]);
                partial[tile] = mm;
            }
        }
    }
    if constexpr (!GEN) {      // regular stencil: keep q = -f/4 (exact), the update is then one fused multiply-add
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = make_float4(-0.25f * f[r].x, -0.25f * f[r].y, -0.25f * f[r].z, -0.25f * f[r].w);
    }
    // general coefficients (compile away when !GEN)
    const float cw0 = (GEN && x + 0 == g.x.n) ? g.x.cw_last : 1.0f, dx0 = (GEN && x + 0 == g.x.n) ? g.x.d_last : 2.0f;
    const float cw1 = (GEN && x + 1 == g.x.n) ? g.x.cw_last : 1.0f, dx1 = (GEN && x + 1 == g.x.n) ? g.x.d_last : 2.0f;
    const float cw2 = (GEN && x + 2 == g.x.n) ? g.x.cw_last : 1.0f, dx2 = (GEN && x + 2 == g.x.n) ? g.x.d_last : 2.0f;
    const float cw3 = (GEN && x + 3 == g.x.n) ? g.x.cw_last : 1.0f, dx3 = (GEN && x + 3 == g.x.n) ? g.x.d_last : 2.0f;
    // reciprocal diagonals, one division per lane and component instead of one per point and half-step:
    // rdA for regular rows (diagonal dx + 2), rdB for the last row (dx + d_last of the row direction)
    const float rdA0 = 1.0f / (dx0 + 2.0f), rdA1 = 1.0f / (dx1 + 2.0f), rdA2 = 1.0f / (dx2 + 2.0f), rdA3 = 1.0f / (dx3 + 2.0f);
    const float rdB0 = GEN ? 1.0f / (dx0 + g.y.d_last) : 0.25f, rdB1 = GEN ? 1.0f / (dx1 + g.y.d_last) : 0.25f;
    const float rdB2 = GEN ? 1.0f / (dx2 + g.y.d_last) : 0.25f, rdB3 = GEN ? 1.0f / (dx3 + g.y.d_last) : 0.25f;
#define SC_TB_GS(L, R_, A, B, FF, CW, K)                                                          \
    (GEN ? ((__builtin_fmaf((CW), (L), (R_)) + __builtin_fmaf(cn, (A), (B))) - (FF)) * (ylast ? rdB##K : rdA##K)        \
         : __builtin_fmaf((((L) + (R_)) + ((A) + (B))), 0.25f, (FF)))     /* regular stencil: FF is -f/4, one rounding as in 0.25 (S - f) */
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
            const bool ylast = GEN && (y == g.y.n);
            const float cn = ylast ? g.y.cw_last : 1.0f;
            (void)cn;
            // x is a multiple of 4 and y0 is even (tile origins and halos are even), so the colour
            // of component k of row r is known at compile time
            if (((r + color) & 1) == 0) {
                float l = wave_from_left(cur.w);
                const float g0 = SC_TB_GS(l, cur.y, a.x, b.x, f[r].x, cw0, 0);
                const float g2 = SC_TB_GS(cur.y, cur.w, a.z, b.z, f[r].z, cw2, 2);
                const float n0 = SOR ? (cur.x + omega * (g0 - cur.x)) : g0;
                const float n2 = SOR ? (cur.z + omega * (g2 - cur.z)) : g2;
                cur.x = (yok & x0ok) ? n0 : cur.x;
                cur.z = (yok & x2ok) ? n2 : cur.z;
            } else {
                float rr = wave_from_right(cur.x);
                const float g1 = SC_TB_GS(cur.x, cur.z, a.y, b.y, f[r].y, cw1, 1);
                const float g3 = SC_TB_GS(cur.z, rr, a.w, b.w, f[r].w, cw3, 3);
                const float n1 = SOR ? (cur.y + omega * (g1 - cur.y)) : g1;
                const float n3 = SOR ? (cur.w + omega * (g3 - cur.w)) : g3;
                cur.y = (yok & x1ok) ? n1 : cur.y;
                cur.w = (yok & x3ok) ? n3 : cur.w;
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
    if (lane < HXQ || lane >= 64 - HXQ || x >= P || x >= W) return;
    float *__restrict__ out = Uout.at(c);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yr = wv * R + r, y = y0 + r;
        if (yr >= HY && yr < RH - HY && y >= 0 && y < H) *reinterpret_cast<float4 *>(out + (size_t)y * P + x) = u[r];
    }
}

#undef SC_TB_GS

// ---------------------------------------------------------------------------- Jacobi
// HXQ = halo float4 lanes per side: 1 (4 columns, T <= 4) or 2 (8 columns, T <= 8)
template <int T, int NW, int R, int TAG, int HXQ>
__global__ __launch_bounds__(NW * 64) void k_jacobi_tb(Field Uin, Field Uout, Field F)
{
    constexpr int HY = T, RH = NW * R, HX = 4 * HXQ;
    static_assert(T <= HX, "column halo too small for this depth");
    __shared__ float4 edge[2][NW][2][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR: row tests become scalar
    const int W = Uin.W, H = Uin.H, P = Uin.pitch;
    // 1-D launch; neighbouring tiles are given to the same XCD so that they share its L2 (sc_wave.h)
    const int nbx = (W + (256 - 2 * HX) - 1) / (256 - 2 * HX), nby = (H + (RH - 2 * HY) - 1) / (RH - 2 * HY);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int bx = tile % nbx, by = (tile / nbx) % nby, c = tile / (nbx * nby);
    const int x = bx * (256 - 2 * HX) - HX + 4 * lane;
    const int ry = by * (RH - 2 * HY) - HY;
    const int y0 = ry + wv * R;
    float4 u[R], f[R];
    tb_load<R>(Uin.at(c), P, H, x, y0, u);
    tb_load<R>(F.at(c), P, H, x, y0, f);
    // The update 0.25 (S - f) is one fused multiply-add on q = -f/4: S/4 and q are exact (powers of two), so the fma's
    // single rounding is that of 0.25 (S - f), bit for bit the subtract-then-scale form.
#pragma unroll
    for (int r = 0; r < R; ++r) f[r] = make_float4(-0.25f * f[r].x, -0.25f * f[r].y, -0.25f * f[r].z, -0.25f * f[r].w);
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
            float l = wave_from_left(cur.w), rr = wave_from_right(cur.x);
            float4 nw = cur;
            const float n0 = __builtin_fmaf((l + cur.y) + (prev.x + b.x), 0.25f, f[r].x);      // f holds -f/4, see above
            const float n1 = __builtin_fmaf((cur.x + cur.z) + (prev.y + b.y), 0.25f, f[r].y);
            const float n2 = __builtin_fmaf((cur.y + cur.w) + (prev.z + b.z), 0.25f, f[r].z);
            const float n3 = __builtin_fmaf((cur.z + rr) + (prev.w + b.w), 0.25f, f[r].w);
            nw.x = (yok & x0ok) ? n0 : nw.x;
            nw.y = (yok & x1ok) ? n1 : nw.y;
            nw.z = (yok & x2ok) ? n2 : nw.z;
            nw.w = (yok & x3ok) ? n3 : nw.w;
            prev = cur;
            u[r] = nw;
        }
        if (step + 1 < T) {
            edge[buf ^ 1][wv][0][lane] = u[0];
            edge[buf ^ 1][wv][1][lane] = u[R - 1];
            __syncthreads();
        }
    }
    if (lane < HXQ || lane >= 64 - HXQ || x >= P || x >= W) return;
    float *__restrict__ out = Uout.at(c);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yr = wv * R + r, y = y0 + r;
        if (yr >= HY && yr < RH - HY && y >= 0 && y < H) *reinterpret_cast<float4 *>(out + (size_t)y * P + x) = u[r];
    }
}

// ---------------------------------------------------------------------------- launchers
constexpr int TB_NW = 8, TB_R = 8;

template <int T, int NW, bool SOR, bool GEN, int FLAGS, int R = TB_R>
static int launch_rb_t(Field Uin, Field Uout, Field F, float omega, const MGGeom &g, Field E, float *partial, hipStream_t s,
                       const RagMember *rag = nullptr, int lev = 0)
{
    constexpr int HXQ = 2 * T <= 4 ? 1 : 2, HX = 4 * HXQ;
    constexpr int RH = NW * R, HY = 2 * T;
    const int blocks = ((Uin.W + (256 - 2 * HX) - 1) / (256 - 2 * HX)) * ((Uin.H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * Uin.C;
    hipLaunchKernelGGL((k_rb_tb<T, NW, R, SOR, GEN, FLAGS, HXQ>), dim3(blocks), dim3(NW * 64), 0, s, Uin, Uout, F, omega, g, E, partial, rag, lev);
    return blocks;
}

bool launch_rb_tb(Field Uin, Field Uout, Field F, int sweeps, float omega, hipStream_t s, bool tag)
{
    const bool sor = omega != 1.0f;
    MGGeom g{};
    Field e{};
    if (tag && !sor) {
        if (sweeps == 1) { launch_rb_t<1, TB_NW, false, false, TB_TAG>(Uin, Uout, F, omega, g, e, nullptr, s); return true; }
        if (sweeps == 2) { launch_rb_t<2, TB_NW, false, false, TB_TAG>(Uin, Uout, F, omega, g, e, nullptr, s); return true; }
        if (sweeps == 4) { launch_rb_t<4, TB_NW, false, false, TB_TAG>(Uin, Uout, F, omega, g, e, nullptr, s); return true; }
    }
    switch (sweeps) {
    case 1: sor ? launch_rb_t<1, TB_NW, true, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s) : launch_rb_t<1, TB_NW, false, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s); return true;
    case 2: sor ? launch_rb_t<2, TB_NW, true, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s) : launch_rb_t<2, TB_NW, false, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s); return true;
    case 3: sor ? launch_rb_t<3, TB_NW, true, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s) : launch_rb_t<3, TB_NW, false, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s); return true;
    case 4: sor ? launch_rb_t<4, TB_NW, true, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s) : launch_rb_t<4, TB_NW, false, false, 0>(Uin, Uout, F, omega, g, e, nullptr, s); return true;
    default: return false;
    }
}

// Level-0 post-smoothing fused with the prolongation of the coarse correction E and the
// max|correction| reduction (Gauss-Seidel, exact level-0 stencil).  Returns the number of
// partial maxima written (0: unsupported depth).
int launch_rb_tb_prolong0(Field Uin, Field Uout, Field F, int sweeps, const MGGeom &g, Field E, float *partial, hipStream_t s)
{
    if (sweeps == 1) return launch_rb_t<1, TB_NW, false, false, TB_PROLONG | TB_MAXC>(Uin, Uout, F, 1.0f, g, E, partial, s);
    if (sweeps == 2) return launch_rb_t<2, TB_NW, false, false, TB_PROLONG | TB_MAXC>(Uin, Uout, F, 1.0f, g, E, partial, s);
    return 0;
}

int tb_blocks_level0(int W, int H, int C, int sweeps)
{
    const int RH = TB_NW * TB_R, HY = 2 * sweeps;
    return ((W + (256 - 2 * TB_HX) - 1) / (256 - 2 * TB_HX)) * ((H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * C;
}

// Side above which a coarse level would use 8-row bands (fewer, larger workgroups).  Measured on
// MI355X (bench.py, 2048^2 and 4096^2 ROIs): 4-row bands win at every coarse-level size -- they
// stay under 128 VGPRs (2 workgroups per CU) and halve the serial work per lane -- so the default
// never selects the 8-row form.
long tb_big_side()
{
    return 1000000L;
}

// coarse multigrid levels: Gauss-Seidel only (omega = 1); smaller workgroups on small levels so
// the grid still spreads over the chip.  mode: 0 plain, TB_ZEROIN, TB_PROLONG (with E).
// Rows per lane (band height) of a coarse-level launch: 8 waves x R rows per workgroup, R in {4, 6, 8}.
// These launches are latency bound (a workgroup's life is a chain of barriers, not bandwidth), and at
// ~100 VGPRs two workgroups fit a CU, so what matters is the number of rounds the grid needs over the
// chip's 512 slots: take the smallest R whose grid fits one round; if none does (large levels) the
// short 4-row bands won every measurement on single clones (tools/bench_configs.py c3/c4).
int tb_gen_rows(int W, int H, int C, int hx, int hy)
{
    if ((long)W * H >= tb_big_side() * tb_big_side()) return 8;
    const int nbx = (W + (256 - 2 * hx) - 1) / (256 - 2 * hx);
    for (int R = 4; R <= 8; R += 2) {
        const int rows = 8 * R - 2 * hy;
        if (nbx * ((H + rows - 1) / rows) * C <= 512) return R;
    }
    // Many rounds either way.  A single clone's large levels: the short 4-row bands (above).  A GROUP of clones (C > 3: tens of
    // rounds, throughput not latency): 6-row bands -- 36 of 48 rows exact at depth 2 instead of 20 of 32 -- measured +0.9 % on the
    // bench step of 32 x 2048^2 (round 4, tools/ab_step.py: 6.263 -> 6.205 ms, three alternating repetitions within 0.1 %).
    return C > 3 ? 6 : 4;
}

// The same choice for launches of depth 3 or 4 (a level that does all its smoothing before the restriction): their halo
// is 2T + 2 = 8 or 10 rows, so 4-row bands keep only 16 or 12 of 32 rows and are worth it only while they make the grid
// fit one round; beyond that 6-row bands (28 of 48 at depth 4) are the efficient form -- 8-row bands spill with the
// general coefficients.  Measured on level 1 of a group of eight 2048^2 clones: 141 us with 4 rows, 103 us with 6.
int tb_gen_rows_deep(int W, int H, int C, int hx, int hy)
{
    const int R = tb_gen_rows(W, H, C, hx, hy);
    if (R != 4) return 6;
    const int nbx = (W + (256 - 2 * hx) - 1) / (256 - 2 * hx), rows = 8 * 4 - 2 * hy;
    return (nbx * ((H + rows - 1) / rows) * C <= 512) ? 4 : 6;
}

// coarse multigrid levels: Gauss-Seidel only (omega = 1).  mode: 0 plain, TB_ZEROIN, TB_PROLONG (with E).
bool launch_rb_tb_gen(Field Uin, Field Uout, Field F, int sweeps, const MGGeom &g, int mode, Field E, hipStream_t s, const RagMember *rag, int lev)
{
    if (sweeps != 1 && sweeps != 2) return false;
    const int R = tb_gen_rows(Uin.W, Uin.H, Uin.C, TB_HX, 2 * sweeps);
    if (rag) {          // a size class: the post-smoothing form of the default schedule (two sweeps behind the prolongation)
        if (sweeps != 2 || mode != TB_PROLONG) return false;
        R == 8 ? launch_rb_t<2, 8, false, true, TB_PROLONG | TB_RAG, 8>(Uin, Uout, F, 1.0f, g, E, nullptr, s, rag, lev)
      : R == 6 ? launch_rb_t<2, 8, false, true, TB_PROLONG | TB_RAG, 6>(Uin, Uout, F, 1.0f, g, E, nullptr, s, rag, lev)
               : launch_rb_t<2, 8, false, true, TB_PROLONG | TB_RAG, 4>(Uin, Uout, F, 1.0f, g, E, nullptr, s, rag, lev);
        return true;
    }
#define SC_GEN_R(TT, MODE)                                                                                          \
    (R == 8 ? launch_rb_t<TT, 8, false, true, MODE, 8>(Uin, Uout, F, 1.0f, g, E, nullptr, s)                        \
   : R == 6 ? launch_rb_t<TT, 8, false, true, MODE, 6>(Uin, Uout, F, 1.0f, g, E, nullptr, s)                        \
            : launch_rb_t<TT, 8, false, true, MODE, 4>(Uin, Uout, F, 1.0f, g, E, nullptr, s))
#define SC_GEN_CASE(TT)                                                                                             \
    { if (mode == 0) SC_GEN_R(TT, 0); else if (mode == TB_ZEROIN) SC_GEN_R(TT, TB_ZEROIN); else SC_GEN_R(TT, TB_PROLONG); }
    if (sweeps == 1) SC_GEN_CASE(1) else SC_GEN_CASE(2)
#undef SC_GEN_CASE
#undef SC_GEN_R
    return true;
}

template <int T>
static void launch_jacobi_t(Field Uin, Field Uout, Field F, hipStream_t s, bool tag)
{
    constexpr int HXQ = T <= 4 ? 1 : 2, HX = 4 * HXQ;
    constexpr int RH = TB_NW * TB_R, HY = T;
    const dim3 grid(((Uin.W + (256 - 2 * HX) - 1) / (256 - 2 * HX)) * ((Uin.H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * Uin.C);
    if (tag) hipLaunchKernelGGL((k_jacobi_tb<T, TB_NW, TB_R, 1, HXQ>), grid, dim3(TB_NW * 64), 0, s, Uin, Uout, F);
    else hipLaunchKernelGGL((k_jacobi_tb<T, TB_NW, TB_R, 0, HXQ>), grid, dim3(TB_NW * 64), 0, s, Uin, Uout, F);
}

bool launch_jacobi_tb(Field Uin, Field Uout, Field F, int sweeps, hipStream_t s, bool tag)
{
    switch (sweeps) {
    case 1: launch_jacobi_t<1>(Uin, Uout, F, s, tag); return true;
    case 2: launch_jacobi_t<2>(Uin, Uout, F, s, tag); return true;
    case 3: launch_jacobi_t<3>(Uin, Uout, F, s, tag); return true;
    case 4: launch_jacobi_t<4>(Uin, Uout, F, s, tag); return true;
    case 6: launch_jacobi_t<6>(Uin, Uout, F, s, tag); return true;
    case 8: launch_jacobi_t<8>(Uin, Uout, F, s, tag); return true;
    default: return false;
    }
}

// default depth (sweeps_per_launch = 0) and the deepest instantiated one
int tb_max_depth(int method) { return method == SC_METHOD_JACOBI ? 8 : 4; }
int tb_hard_max_depth(int method) { return method == SC_METHOD_JACOBI ? 8 : 4; }

} // namespace sc
