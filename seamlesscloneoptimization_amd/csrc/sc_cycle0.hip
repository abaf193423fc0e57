// sc_cycle0.hip -- the whole level-0 part of a multigrid V-cycle in ONE launch (gfx950).
//
// k_cycle0<T, NW, R, PRO, GEN, ZEROIN, TAG> reads U and the RHS once and, with the field held in
// registers (same register-blocked layout as sc_sweep_tb.hip), performs
//     [PRO]  U += P*E  (bilinear prolongation of the coarse correction) and the per-block
//            max|P*E| used by the stop rule,
//     T      red-black Gauss-Seidel sweeps (post-smoothing of this cycle and pre-smoothing of the
//            next one back to back: T = post + pre; the first launch of a solve has T = pre; the
//            cycle the stop rule judges has T = post and stops here: TAG bit 3),
//     then   the residual (float32 in difference form on level 0, see below) and its full-weighting
//            restriction to the coarse RHS,
// and writes U and the coarse RHS once.  Against the three-kernel form (smoother, residual +
// restriction, prolongation + smoother: ~39 B of fabric traffic per unknown and cycle) this moves
// ~12 B.  Level 0 (GEN = false): exact 5-point stencil, regular spacing, the restriction is the plain
// 1/4-1/2-1/4 tensor stencil up to a normalisation factor at the last coarse row/column
// (MGDim::inv_last); the RHS and, for the first launch, the incoming field may be float16 where the
// pre-process stored them so (TAG bits 1, 2).  GEN = true: the same kernel on a coarse level, general
// coefficients at the last column / row, starting from a zero correction (ZEROIN).
//
// Workgroups are numbered so that neighbouring tiles share an XCD and its L2 (xcd_tile, sc_wave.h).
//
// On level 0 a wave that lies strictly inside the domain (a wave-uniform test, three waves in four of
// a large ROI) takes forms of the prolongation, the half-steps and the residual without masks, ghost
// values or validity selects; they produce the same values, operation for operation, as the checked
// forms.  Weights that are powers of two are applied inside fused multiply-adds (exact products: the
// single rounding is the one the separate multiply-and-add forms perform).
//
// Halo: values at depth d from the region edge are exact for d half-steps; the residual needs one
// more ring and the restriction a second one, so the exact output tile is the region minus
// HX = 12 columns (3 float4 lanes) and HY = 2T+2 rows on every side.  HY is even so that coarse
// rows (even fine rows) pair up inside each lane's R-row band.
#include "sc_common.h"
#include <hip/hip_fp16.h>
#include "sc_wave.h"
#include "sc_mg_device.h"
#include <type_traits>

namespace sc {

constexpr int C0_HXQ = 3;               // halo lanes (float4) per side
constexpr int C0_HX = 4 * C0_HXQ;

// R rows of one lane's float4 column.  Addresses are clamped into the plane instead of tested: a
// test around each load would put it in its own branch region with a wait behind it, and the R
// rows would be fetched one memory latency after the other instead of all at once.  What a clamped
// lane / row receives is never used: the ring is fixed and every update and residual is masked to
// the interior, so out-of-range values only ever flow into other out-of-range values.
template <int R>
__device__ __forceinline__ void c0_load(const float *__restrict__ p, int P, int H, int x, int y0, float4 (&v)[R])
{
    const int xc = min(max(x, 0), P - 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);
        v[r] = *reinterpret_cast<const float4 *>(p + (size_t)yc * P + xc);
    }
}

// The level-0 right-hand side of a clone is an integer in [-1020, 1020] (sums of four differences of 8-bit values,
// blended with a 0/1 mask), which float16 holds exactly; the pre-process can store it as such (sc_kernels.hip) and
// the cycle kernel then reads 2 bytes per unknown instead of 4.  Same element pitch / plane size as the float field.
template <int R>
__device__ __forceinline__ void c0_load_half(const __half *__restrict__ p, int P, int H, int x, int y0, float4 (&v)[R])
{
    const int xc = min(max(x, 0), P - 4);
    uint2 raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);
        raw[r] = *reinterpret_cast<const uint2 *>(p + (size_t)yc * P + xc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const __half2 a = *reinterpret_cast<const __half2 *>(&raw[r].x), b = *reinterpret_cast<const __half2 *>(&raw[r].y);
        const float2 fa = __half22float2(a), fb = __half22float2(b);
        v[r] = make_float4(fa.x, fa.y, fb.x, fb.y);
    }
}

// The same rows kept as they are stored: four float16 values in two registers per row.  The sweeps and the residual
// subtract them with v_fma_mix_f32, which widens a float16 operand on the fly (exact), so the right-hand side costs the
// kernel 2R registers instead of 4R and no conversion instructions.
template <int R>
__device__ __forceinline__ void c0_load_half_raw(const __half *__restrict__ p, int P, int H, int x, int y0, uint2 (&v)[R])
{
    const int xc = min(max(x, 0), P - 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);
        v[r] = *reinterpret_cast<const uint2 *>(p + (size_t)yc * P + xc);
    }
}
// The field BETWEEN the level-0 launches of the fast path as 16-bit fixed point (TAG bits 8, 9): code = trunc(64 u + 16384.5)
// clamped to [0, 65535], u = code / 64 - 256: the range [-256, 768) in steps of 1/64.  With a CONSERVATIVE guidance field (the
// gradient of one image: a mask that is all 255 inside its bounding box) the solution of a clone lies in [-255, 510] -- it is
// the source patch plus a discrete harmonic function whose boundary values are differences of 8-bit values.  A mask that mixes
// patch and destination gradients pixel by pixel (holes, stripes, rings) makes the field non-conservative, and then neither the
// solution nor the iterates are bounded by the images' range (rings of inward ramps pile up 1500 grey levels at 512^2).  So the
// range is CHECKED: a store that saturates reports itself (c0_q16_checked, AbortFlag), the output launches of that solve then
// write nothing and the host repeats the clone on float fields (sc_run_info.field_retry).  8-bit boundary values are exact,
// and a rounding of at most 1/128 per stored value is to a multigrid iterate what one more high-frequency error component is:
// the cycles that follow remove it like any other error.  Only the first stores of a solve use the format (sc_multigrid.cpp:
// the launch before the judged cycle writes float again, so two cycles lie between the last rounding and the output).  Same
// element pitch / plane size as the float field, 2 bytes per unknown instead of 4.
template <int R>
__device__ __forceinline__ void c0_load_q16(const uint16_t *__restrict__ p, int P, int H, int x, int y0, float4 (&v)[R])
{
    const int xc = min(max(x, 0), P - 4);
    uint2 raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yc = min(max(y0 + r, 0), H - 1);
        raw[r] = *reinterpret_cast<const uint2 *>(p + (size_t)yc * P + xc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        v[r].x = __builtin_fmaf((float)(raw[r].x & 0xffffu), 0.015625f, -256.0f);
        v[r].y = __builtin_fmaf((float)(raw[r].x >> 16), 0.015625f, -256.0f);
        v[r].z = __builtin_fmaf((float)(raw[r].y & 0xffffu), 0.015625f, -256.0f);
        v[r].w = __builtin_fmaf((float)(raw[r].y >> 16), 0.015625f, -256.0f);
    }
}
// code of u, and `seen |= the unclamped integer`: any bit above the 16th in `seen` (a negative value sets the sign bits) says a
// value left the range.  v_cvt_i32_f32 saturates, so there is no wrap-around for wild values; same codes as
// trunc(clamp(64 u + 16384.5, 0, 65535)) bit for bit.
__device__ __forceinline__ unsigned c0_q16_checked(float u, int &seen)
{
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(__builtin_fmaf(u, 64.0f, 16384.5f)));
    seen |= i;
    return (unsigned)min(max(i, 0), 65535);
}

// Level 0 (regular stencil) holds q = -f/4 instead of f (exact in either format: a power-of-two scaling): the Gauss-Seidel
// update 0.25 (S - f) is then ONE fused multiply-add, fma(S, 0.25, q) -- S/4 and q are exact, so the single rounding of the
// fma is the rounding of 0.25 (S - f), bit for bit what the subtract-then-scale form gives -- and the residual's f - X is
// fma(q, -4, -X), again one rounding.
__device__ __forceinline__ void c0_scale_q(uint2 &h)
{
    const __half2 k = __float2half2_rn(-0.25f);
    __half2 a = *reinterpret_cast<__half2 *>(&h.x), b = *reinterpret_cast<__half2 *>(&h.y);
    a = __hmul2(a, k); b = __hmul2(b, k);
    h.x = *reinterpret_cast<unsigned *>(&a); h.y = *reinterpret_cast<unsigned *>(&b);
}
template <int K>
__device__ __forceinline__ float c0_gs_q(float s, float quarter, const uint2 &h)      // 0.25 s + q[K]
{
    const unsigned raw = (K < 2) ? h.x : h.y;
    float d;
    if (K & 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(s), "s"(quarter), "v"(raw));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(s), "s"(quarter), "v"(raw));
    return d;
}
template <int K>
__device__ __forceinline__ float c0_f_minus(const uint2 &h, float s)      // f[K] - s = -4 q[K] - s
{
    const unsigned raw = (K < 2) ? h.x : h.y;
    float d;
    if (K & 1) asm("v_fma_mix_f32 %0, %1, -4.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(raw), "v"(s));
    else asm("v_fma_mix_f32 %0, %1, -4.0, -%2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(raw), "v"(s));
    return d;
}
__device__ __forceinline__ float c0_comp(const float4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// TAG bit 4 (level 0 only): level 1 runs without post-smoothing and without its own prolongation launch.  Its finished
// correction E1 = U1 + P21 E2 (U1: level-1 correction after pre-smoothing, E2: finished level-2 correction) is formed on
// the fly where this kernel interpolates from it: two extra level-2 values per level-2 row and a few adds per lane.
// Leaving out the level-1 post-smoothing changes the contraction per cycle by a few percent (oracle/mg_np.py carries
// the same schedule); leaving out its launch is worth ~10 % of the clone throughput.
// (ComposeArgs: sc_common.h.  The prolongation below is the level-0 specialisation of sc_mg_device.h's prolong_apply -- no
// LEFT column, level 0 has at most one tail point; kept inline here because the shared form costs this kernel 11 VGPRs and
// ~100 SGPR spill moves: 48.7 instead of 45.6 us per launch.)

// GEN    = coarse multigrid level: general stencil coefficients at the last column / row (MGDim)
//          and the interpolation-tail weights in the restriction of the last coarse column / row.
// ZEROIN = the incoming correction is identically zero and is not read (first visit of a level).
// TAG bit 0: second symbol for isolated timing; bit 1: F is float16; bit 2: Uin is float16; bit 3: last cycle (no residual /
// restriction); bit 4: composed prolongation; bit 5: last cycle whose result leaves as bytes (Uout's memory receives planar
// 8-bit output values: float-table node correction added, clamped, truncated) instead of as a field; bit 6: the launch
// leaves the float-table correction's cell shares of the field it writes in `bands` (the last-cycle form does so whenever
// `bands` is not null)
// TAG bit 10 (round 5): the launch serves a SIZE CLASS (RagMember, sc_common.h) -- the fields' strides and the grid are the class's,
// everything else (field size, this level's and the next one's geometry, the coarse planes' row counts) is the member's, read from
// rag[channel / 3] (a scalar load) at entry; `lev` = the level of Uin / F in the member's hierarchy.  Tiles beyond the member's
// extent leave (their partial maximum is 0).
template <int T, int NW, int R, bool PRO, bool GEN, bool ZEROIN, int TAG = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4))) void k_cycle0(Field Uin, Field Uout, Field F, Field Fc, Field E, MGGeom g,
                                                    float *__restrict__ partial, ComposeArgs comp, float4 *__restrict__ bands, LmNodes lm, AbortFlag sat,
                                                    const RagMember *__restrict__ rag, int lev)
{
    constexpr int HY = 2 * T + 2, RH = NW * R;
    constexpr bool RAG = (TAG & 1024) != 0;
    constexpr bool COMP = (TAG & 16) != 0;      // E is U1; the interpolated level-2 correction is added on the fly (ComposeArgs)
    static_assert(!COMP || (PRO && !GEN && R % 2 == 0), "composition of two prolongations exists on level 0 only");
    constexpr bool FINAL = (TAG & 8) != 0;      // prolongation + post-smoothing only: the cycle the stop rule is expected to accept
    constexpr bool OUT = (TAG & 32) != 0, BANDS = ((TAG & 64) != 0 || FINAL) && !OUT;
    static_assert(!OUT || (FINAL && !GEN), "bytes leave from the last level-0 launch only");
    static_assert(!(TAG & 64) || !GEN, "cell shares exist on level 0 only");
    constexpr bool HF = (TAG & 2) != 0, HU = (TAG & 4) != 0;   // HU: the first launch of a clone reads the 8-bit destination values the pre-process stored as float16
    static_assert(!(HF && GEN), "float16 right-hand sides exist on level 0 only");
    // TAG bit 7: LEVEL 1's right-hand side and correction are stored as float16 (same element pitch / plane size inside their
    // float buffers).  On level 0 (!GEN) that is the restriction this launch writes and the correction it interpolates from; on
    // the level-1 launch itself (GEN, ZEROIN) its own F and the correction it writes.  A correction scheme does not care: the
    // coarse problem is solved to a factor 0.05 per cycle anyway, a relative 5e-4 on its data moves the iterates by that much
    // of a correction and not the fixed point (level 0's residual is exact).  Halves the traffic of the level-1 launch and takes
    // 1 byte per unknown off every level-0 launch (oracle/mg_np.py rounds the same two fields).
    constexpr bool L1H = (TAG & 128) != 0;
    // TAG bits 8 / 9: the incoming / outgoing field is 16-bit fixed point (c0_load_q16)
    constexpr bool UQI = (TAG & 256) != 0, UQO = (TAG & 512) != 0;
    static_assert(!((UQI || UQO) && (GEN || (HU && UQI) || (OUT && UQO))), "16-bit fields: level 0, between its launches");
    static_assert(!L1H || !GEN || ZEROIN, "float16 level-1 fields: the level-1 launch starts from a zero correction");
    static_assert(!(PRO && GEN), "the in-kernel prolongation relies on level 0's regular last interval");
    static_assert(2 * T + 2 <= C0_HX, "column halo too small");
    static_assert(R % 2 == 0, "bands must hold whole coarse-row pairs");
    __shared__ float4 edge[2][NW][2][64];
    __shared__ float2 hedge[2][NW][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR: row tests become scalar
    int W = Uin.W, H = Uin.H;
    const int P = Uin.pitch;
    // 1-D launch; the tile this workgroup owns is chosen so that neighbouring tiles share an XCD (L2)
    const int nbx = (W + (256 - 2 * C0_HX) - 1) / (256 - 2 * C0_HX), nby = (H + (RH - 2 * HY) - 1) / (RH - 2 * HY);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int bx = tile % nbx, by = (tile / nbx) % nby, c = tile / (nbx * nby);
    if constexpr (RAG) {
        const RagMember &m = rag[c / 3];
        W = m.lw[lev]; H = m.lh[lev];
        if (bx * (256 - 2 * C0_HX) >= W || by * (RH - 2 * HY) >= H) {      // nothing of this member in the tile's exact output (block-uniform, before any barrier)
            if (PRO && threadIdx.x == 0) partial[tile] = 0.f;
            return;
        }
        g = m.g[lev];
        if (PRO) E.H = m.lh[lev + 1];
        if ((TAG & 16) != 0) { comp.g1 = m.g[lev + 1]; comp.E2.H = m.lh[lev + 2]; }
    }
    const int x = bx * (256 - 2 * C0_HX) - C0_HX + 4 * lane;
    const int y0 = by * (RH - 2 * HY) - HY + wv * R;       // even
    float4 u[R], f[HF ? 1 : R];
    uint2 fh[HF ? R : 1];        // float16 right-hand side, kept packed (c0_minus_f)
    if (ZEROIN) {
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (HU) {
        c0_load_half<R>(reinterpret_cast<const __half *>(Uin.p) + (size_t)c * Uin.plane, P, H, x, y0, u);
    } else if (UQI) {
        c0_load_q16<R>(reinterpret_cast<const uint16_t *>(Uin.p) + (size_t)c * Uin.plane, P, H, x, y0, u);
    } else {
        c0_load<R>(Uin.at(c), P, H, x, y0, u);
    }
    // Coarse values the lane interpolates from, requested together with U: level-1 rows y0/2 - 1 .. y0/2 + R/2 at columns
    // x/2 .. x/2 + 2 (the extra row above feeds the ghost row, see the prolongation), and for the composed form the
    // level-2 rows under them at columns x/4, x/4 + 1.  Indices are clamped; what a clamped index reads is a ring or pad
    // value (zero) or is never used.
    constexpr int NE = R / 2 + 2, NQ = R / 4 + 3;
    float2 eab[PRO ? NE : 1];
    float ecc[PRO ? NE : 1];
    float2 e2r[COMP ? NQ : 1];
    float e2l[COMP ? NQ : 1];          // level-2 column x/4 - 1: source of the ghost value when x/4 itself is the ghost column
    // Round 4: a lane loads only ITS OWN coarse columns -- level 1's x/2, x/2 + 1 (one 4- or 8-byte load per row) and level 2's x/4
    // (one load per row) -- and takes column x/2 + 2 (= the right lane's x/2), x/4 + 1 and x/4 - 1 from the neighbouring lanes with one
    // full-wave DPP shift each, after the loads have landed (PRO_SHARE below): 11 loads per lane instead of 27.  The wave's two end
    // lanes have no neighbour: lane 63 fetches its level-2 column x/4 + 1 itself (its level-1 column x/2 + 2 only feeds the wave's very
    // last fine column, which the column halo absorbs: 2T + 2 + 1 <= C0_HX, asserted below), lane 0's level-2 column x/4 - 1 only
    // matters where x/4 is the ghost column of the domain's right end, two columns into a halo lane.  Where an address is clamped
    // (lanes outside the domain) neighbours need not agree; those values are masked or never reach an exact output, as before.
    static_assert(!PRO || GEN || 2 * T + 3 <= C0_HX, "column halo too small for the shared prolongation loads");
    float e2b63[COMP ? NQ : 1];        // lane 63's own level-2 column x/4 + 1
    if (PRO) {
        const int cx = min(max(x >> 1, 0), E.pitch - 4), J = (y0 >> 1) - 1;
        if constexpr (L1H) {
            const __half *__restrict__ e = reinterpret_cast<const __half *>(E.p) + (size_t)c * E.plane;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const __half *er = e + (size_t)min(max(J + j, 0), E.H - 1) * E.pitch + cx;      // cx is even: a 4-byte aligned pair
                eab[j] = __half22float2(*reinterpret_cast<const __half2 *>(er));
            }
        } else {
        const float *__restrict__ e = E.at(c);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const float *er = e + (size_t)min(max(J + j, 0), E.H - 1) * E.pitch + cx;
            eab[j] = *reinterpret_cast<const float2 *>(er);
        }
        }
    }
    if (COMP) {
        const float *__restrict__ e2 = comp.E2.at(c);
        const int qx = min(max(x >> 2, 0), comp.E2.pitch - 2), Q = ((y0 >> 1) - 1) >> 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float *er = e2 + (size_t)min(max(Q + q, 0), comp.E2.H - 1) * comp.E2.pitch + qx;
            e2r[q].x = er[0];
            e2b63[q] = 0.f;
            if (lane == 63) e2b63[q] = er[1];
        }
    }
    if (!PRO) {   // with PRO the RHS is fetched after the prolongation
        if constexpr (HF) c0_load_half_raw<R>(reinterpret_cast<const __half *>(F.p) + (size_t)c * F.plane, P, H, x, y0, fh);
        else if constexpr (GEN && L1H) c0_load_half<R>(reinterpret_cast<const __half *>(F.p) + (size_t)c * F.plane, P, H, x, y0, f);
        else c0_load<R>(F.at(c), P, H, x, y0, f);
    }
    const bool x0ok = (x + 0 >= 1) && (x + 0 <= W - 2), x1ok = (x + 1 >= 1) && (x + 1 <= W - 2);
    const bool x2ok = (x + 2 >= 1) && (x + 2 <= W - 2), x3ok = (x + 3 >= 1) && (x + 3 <= W - 2);

    // ------------------------------------------------------------------ prolongation
    // Bilinear interpolation from coarse columns x/2, x/2+1, x/2+2 and rows y0/2 .. y0/2+R/2.  The last interval of a level
    // is irregular (MGDim): the one or two fine points beyond the last coarse point nc interpolate between E[nc] and the
    // boundary value 0 with weights tw1 / tw2.  Both equal the REGULAR formula applied to a ghost value
    //     E[nc + 1] := (2 tw1 - 1) E[nc]
    // (tw2 = 2 tw1 - 1: the two tail points lie on one straight line to the boundary), so one code path serves every lane:
    // the ghost column / row is patched in where the lane's window contains index nc + 1.  The same holds one level up for
    // the composed form (level-2 ghosts with level 1's tail weights).
    // Most waves of a launch lie strictly inside the domain on every level involved: all 256 columns and R rows interior,
    // the coarse windows clear of the last coarse column / row (no ghost value, nothing outside the coarse interior).  Those
    // waves take mask-free forms of the prolongation, the half-steps and the residual below; the test is wave-uniform and
    // the values they produce are the same, operation for operation, as the checked forms produce there.
    const int xw = bx * (256 - 2 * C0_HX) - C0_HX;             // column of lane 0
    // (Level 0 only.  The same split on the coarse levels -- regular stencil for waves clear of the last column / row --
    // was measured and changes nothing there: 20.6 vs 20.8 us on level 1 of a 2048^2 ROI; those launches are bounded by the
    // latency of one workgroup, not by instruction count; the many-round launches of a group of clones do not gain either:
    // 104.6 vs 103 us on level 1 of eight 2048^2 clones, level 2 and a single 8192^2 clone get slower.)
    bool inner = !GEN && (xw >= 4) && (xw + 255 <= W - 2) && (y0 >= 1) && (y0 + R - 1 <= H - 2);
    if (PRO) {
        const int Jb = (y0 >> 1) - 1;
        inner = inner && (((xw + 252) >> 1) + 2 <= g.x.nc) && (Jb >= 1) && (Jb + NE - 1 <= g.y.nc);
        if (COMP) inner = inner && (((xw + 252) >> 2) + 1 <= comp.g1.x.nc) && ((Jb >> 1) + NQ - 1 <= comp.g1.y.nc);
    }
    // parity of the first coarse row of the window: with R a multiple of 4 and an even tile step in coarse rows it is T's
    constexpr bool JKNOWN = (R % 4 == 0) && ((RH / 2 - HY) % 2 == 0);
    if (PRO) {
        // PRO_SHARE: the neighbours' columns (see the loads above)
#pragma unroll
        for (int j = 0; j < NE; ++j) ecc[j] = wave_from_right(eab[j].x);
        if (COMP) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float own = e2r[q].x;
                const float fr = wave_from_right(own), fl = wave_from_left(own);      // full-wave shifts: outside any lane-dependent branch
                e2r[q].y = lane == 63 ? e2b63[q] : fr;
                e2l[q] = fl;
            }
        }
        float m = 0.f;
        auto prolong = [&](auto checked) {
            constexpr bool CK = decltype(checked)::value;
            const int ncx = g.x.nc, ncy = g.y.nc;
            const float gx = 2.0f * g.x.tw1 - 1.0f, gy = 2.0f * g.y.tw1 - 1.0f;
            const int c0 = x >> 1, Jb = (y0 >> 1) - 1;                      // first coarse column / first loaded coarse row
            // level-2 row q at level-1 columns c0, c0 + 1 (sum of two, not yet normalised), c0 + 2
            float h0[COMP ? NQ : 1], hs[COMP ? NQ : 1], h2[COMP ? NQ : 1];
            const int Qb = Jb >> 1;
            if (COMP) {
                const int n2x = comp.g1.x.nc, n2y = comp.g1.y.nc, q0 = x >> 2;
                const float g1x = 2.0f * comp.g1.x.tw1 - 1.0f, g1y = 2.0f * comp.g1.y.tw1 - 1.0f;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float a2 = e2r[q].x, b2 = e2r[q].y;
                    if (CK) {
                        a2 = (q0 <= n2x) ? a2 : 0.f; b2 = (q0 + 1 <= n2x) ? b2 : 0.f;
                        if (q0 == n2x) b2 = g1x * a2;                       // ghost column of level 2
                        if (q0 == n2x + 1) a2 = g1x * e2l[q];
                    }
                    h0[q] = a2; hs[q] = a2 + b2; h2[q] = b2;
                }
                if (CK) {
#pragma unroll
                    for (int q = 1; q < NQ; ++q)
                        if (Qb + q == n2y + 1) { h0[q] = g1y * h0[q - 1]; hs[q] = g1y * hs[q - 1]; h2[q] = g1y * h2[q - 1]; }   // ghost row
                }
            }
            // Coarse row j of the window at the lane's four fine columns, NOT yet normalised: (ea, ea + eb, eb, eb + ec).  The
            // weights 1/2 and 1/4 are applied where a value is added, as one fused multiply-add each: a power-of-two weight
            // commutes with every rounding, so fma(t, 1/2, u) is bit for bit u + (0.5 a + 0.5 b).
            // Coarse values outside the coarse interior are zero (ring, pad, or set so here), once per coarse value, instead
            // of masking the fine points: level 0 has a last interval of exactly one spacing, so its ghost value is 0 (odd n)
            // or -E[nc] (even n), and the regular formula then yields exactly 0 on the whole ring; points beyond the ring
            // receive values that never reach the interior and whose magnitudes also occur inside it (same maximum).
            // Rows are produced and consumed in order (coarse row j feeds fine rows 2j - 3 and 2j - 2 of the band), which keeps
            // two coarse rows live instead of the whole window.
            float4 Sp = make_float4(0.f, 0.f, 0.f, 0.f);
            float3 prev = make_float3(0.f, 0.f, 0.f);
            float m1 = 0.f, m2 = 0.f, m4 = 0.f;      // max |t| of the values that enter with weight 1, 1/2, 1/4
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int Jr = Jb + j;
                float ea = eab[j].x, eb = eab[j].y, ec = ecc[j];   // outside the coarse interior: ring or pad, i.e. zero (the level planes' pitch is at least W + 2: no lane that owns an interior point had its column clamped)
                if (COMP) {      // level-1 correction = pre-smoothed U1 + interpolated level-2 correction (P21 E2), zero outside level 1
                    if (JKNOWN) {
                        constexpr int JP = T & 1;
                        const int q = (j + JP) >> 1;                        // level-2 row at / above level-1 row Jb + j
                        if ((j + JP) & 1) {
                            ea = __builtin_fmaf(h0[q] + h0[q + 1], 0.5f, ea); eb = __builtin_fmaf(hs[q] + hs[q + 1], 0.25f, eb);
                            ec = __builtin_fmaf(h2[q] + h2[q + 1], 0.5f, ec);
                        } else { ea = ea + h0[q]; eb = __builtin_fmaf(hs[q], 0.5f, eb); ec = ec + h2[q]; }
                    } else {
                        const int q = (Jr >> 1) - Qb, q1 = q + (Jr & 1);    // even rows: the row itself twice (doubling is exact)
                        ea = __builtin_fmaf(h0[q] + h0[q1], 0.5f, ea); eb = __builtin_fmaf(hs[q] + hs[q1], 0.25f, eb);
                        ec = __builtin_fmaf(h2[q] + h2[q1], 0.5f, ec);
                    }
                    if (CK) {
                        const bool rin = (Jr >= 1) && (Jr <= ncy);
                        ea = (rin && c0 >= 1 && c0 <= ncx) ? ea : 0.f;
                        eb = (rin && c0 + 1 <= ncx) ? eb : 0.f;
                        ec = (rin && c0 + 2 <= ncx) ? ec : 0.f;
                    }
                }
                if (CK) {
                    if (c0 == ncx) eb = gx * ea;                            // ghost column of this level
                    if (c0 + 1 == ncx) ec = gx * eb;
                    if (Jr == ncy + 1) { ea = gy * prev.x; eb = gy * prev.y; ec = gy * prev.z; }   // ghost row
                    prev = make_float3(ea, eb, ec);
                }
                const float4 Sj = make_float4(ea, ea + eb, eb, eb + ec);
                if (j >= 2) {            // odd fine row between coarse rows j - 1 and j
                    const int r = 2 * j - 3;
                    const float4 t = make_float4(Sp.x + Sj.x, Sp.y + Sj.y, Sp.z + Sj.z, Sp.w + Sj.w);
                    u[r].x = __builtin_fmaf(t.x, 0.5f, u[r].x); u[r].y = __builtin_fmaf(t.y, 0.25f, u[r].y);
                    u[r].z = __builtin_fmaf(t.z, 0.5f, u[r].z); u[r].w = __builtin_fmaf(t.w, 0.25f, u[r].w);
                    m2 = fmaxf(fmaxf(m2, fabsf(t.x)), fabsf(t.z));
                    m4 = fmaxf(fmaxf(m4, fabsf(t.y)), fabsf(t.w));
                }
                if (j >= 1 && j < NE - 1) {   // even fine row on coarse row j
                    const int r = 2 * j - 2;
                    u[r].x = u[r].x + Sj.x; u[r].y = __builtin_fmaf(Sj.y, 0.5f, u[r].y);
                    u[r].z = u[r].z + Sj.z; u[r].w = __builtin_fmaf(Sj.w, 0.5f, u[r].w);
                    m1 = fmaxf(fmaxf(m1, fabsf(Sj.x)), fabsf(Sj.z));
                    m2 = fmaxf(fmaxf(m2, fabsf(Sj.y)), fabsf(Sj.w));
                }
                Sp = Sj;
            }
            m = fmaxf(m1, fmaxf(0.5f * m2, 0.25f * m4));
        };
        if (inner) prolong(std::false_type{});
        else if (x >= 0 && x <= W - 2 && y0 + R - 1 >= 1 && y0 <= H - 2) prolong(std::true_type{});   // the lane owns at least one interior point
        // after the prolongation (VGPR pressure), in flight during the reduction
        if constexpr (HF) c0_load_half_raw<R>(reinterpret_cast<const __half *>(F.p) + (size_t)c * F.plane, P, H, x, y0, fh);
        else c0_load<R>(F.at(c), P, H, x, y0, f);
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

    // ------------------------------------------------------------------ T red-black sweeps
    if constexpr (!GEN) {          // level 0 keeps q = -f/4 (c0_gs_q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (HF) c0_scale_q(fh[r]);
            else f[r] = make_float4(-0.25f * f[r].x, -0.25f * f[r].y, -0.25f * f[r].z, -0.25f * f[r].w);
        }
    }
    const float quarter = 0.25f;
    (void)quarter;
    // general coefficients (compile away when !GEN)
    float cw0 = (GEN && x + 0 == g.x.n) ? g.x.cw_last : 1.0f, dx0 = (GEN && x + 0 == g.x.n) ? g.x.d_last : 2.0f;
    float cw1 = (GEN && x + 1 == g.x.n) ? g.x.cw_last : 1.0f, dx1 = (GEN && x + 1 == g.x.n) ? g.x.d_last : 2.0f;
    float cw2 = (GEN && x + 2 == g.x.n) ? g.x.cw_last : 1.0f, dx2 = (GEN && x + 2 == g.x.n) ? g.x.d_last : 2.0f;
    float cw3 = (GEN && x + 3 == g.x.n) ? g.x.cw_last : 1.0f, dx3 = (GEN && x + 3 == g.x.n) ? g.x.d_last : 2.0f;
    // reciprocal diagonals, one division per lane and component instead of one per point and half-step:
    // rdA for regular rows (diagonal dx + 2), rdB for the last row (dx + d_last of the row direction)
    float rdA0 = 1.0f / (dx0 + 2.0f), rdA1 = 1.0f / (dx1 + 2.0f), rdA2 = 1.0f / (dx2 + 2.0f), rdA3 = 1.0f / (dx3 + 2.0f);
    float rdB0 = GEN ? 1.0f / (dx0 + g.y.d_last) : 0.25f, rdB1 = GEN ? 1.0f / (dx1 + g.y.d_last) : 0.25f;
    float rdB2 = GEN ? 1.0f / (dx2 + g.y.d_last) : 0.25f, rdB3 = GEN ? 1.0f / (dx3 + g.y.d_last) : 0.25f;
    if constexpr (GEN) {   // computed once: keep the compiler from re-deriving the divisions inside every checked half-step
        asm volatile("" : "+v"(cw0), "+v"(cw1), "+v"(cw2), "+v"(cw3), "+v"(rdA0), "+v"(rdA1), "+v"(rdA2), "+v"(rdA3));
        asm volatile("" : "+v"(rdB0), "+v"(rdB1), "+v"(rdB2), "+v"(rdB3));
    }
#define SC_C0_F(K) c0_comp(f[HF ? 0 : r], K)
#define SC_C0_GS(G, L, R_, A, B, CW, K)                                                           \
    ((G) ? ((__builtin_fmaf((CW), (L), (R_)) + __builtin_fmaf(cn, (A), (B))) - SC_C0_F(K)) * (ylast ? rdB##K : rdA##K)  \
     : HF ? c0_gs_q<K>((((L) + (R_)) + ((A) + (B))), quarter, fh[HF ? r : 0])                    \
          : __builtin_fmaf((((L) + (R_)) + ((A) + (B))), 0.25f, SC_C0_F(K)))
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    edge[0][wv][0][lane] = u[0];
    edge[0][wv][1][lane] = u[R - 1];
    __syncthreads();
    // inner waves (see above): the half-steps are pure arithmetic, 3 adds + 1 fma per point
#define SC_C0_ROWS(MASKED)                                                                            \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                   \
        const int y = y0 + r;                                                                         \
        const bool yok = (y >= 1) && (y <= H - 2);                                                    \
        const float4 a = (r == 0) ? up : u[r - 1];                                                    \
        const float4 b = (r == R - 1) ? dn : u[r + 1];                                                \
        float4 cur = u[r];                                                                            \
        const bool ylast = GEN && (y == g.y.n);                                                       \
        const float cn = ylast ? g.y.cw_last : 1.0f;                                                  \
        (void)cn; (void)yok;                                                                          \
        if (((r + color) & 1) == 0) {      /* compile time: x is a multiple of 4 and y0 is even */    \
            float l = wave_from_left(cur.w);                                                          \
            const float n0 = SC_C0_GS(GEN, l, cur.y, a.x, b.x, cw0, 0);                                    \
            const float n2 = SC_C0_GS(GEN, cur.y, cur.w, a.z, b.z, cw2, 2);                                \
            cur.x = (!(MASKED) || (yok & x0ok)) ? n0 : cur.x;                                         \
            cur.z = (!(MASKED) || (yok & x2ok)) ? n2 : cur.z;                                         \
        } else {                                                                                      \
            float rr = wave_from_right(cur.x);                                                        \
            const float n1 = SC_C0_GS(GEN, cur.x, cur.z, a.y, b.y, cw1, 1);                                \
            const float n3 = SC_C0_GS(GEN, cur.z, rr, a.w, b.w, cw3, 3);                                   \
            cur.y = (!(MASKED) || (yok & x1ok)) ? n1 : cur.y;                                         \
            cur.w = (!(MASKED) || (yok & x3ok)) ? n3 : cur.w;                                         \
        }                                                                                             \
        u[r] = cur;                                                                                   \
    }
#pragma unroll
    for (int step = 0; step < 2 * T; ++step) {
        const int buf = step & 1, color = step & 1;
        const float4 up = (wv > 0) ? edge[buf][wv - 1][1][lane] : zero;
        const float4 dn = (wv < NW - 1) ? edge[buf][wv + 1][0][lane] : zero;
        if (inner) { SC_C0_ROWS(false) } else { SC_C0_ROWS(true) }
        edge[buf ^ 1][wv][0][lane] = u[0];
        edge[buf ^ 1][wv][1][lane] = u[R - 1];
        __syncthreads();
    }
#undef SC_C0_ROWS

    // ------------------------------------------------------------------ cell shares for the float-table correction
    // (level 0, when the caller asks: `bands`; the last-cycle form and the forms with TAG bit 6).  The correction's restriction (sc_lowmode.hip: every 8 x 8 cell of
    // the finished field sends hat-weighted sums to its four corner nodes) needs one more pass over U; the finished field is
    // in registers right here.  A lane's four columns are half of one cell, its band of R = 8 rows (y0 is even, never a
    // multiple of 8) spans two cell rows: each lane pair writes two cell shares per band, part A for cell row y0 >> 3 and
    // part B for the next; k_lm_bands_to_cells adds the (at most four) parts of a cell in a fixed order.  Rows outside the
    // tile's exact output and everything outside the interior count as zero, exactly as in k_lm_restrict.
    if constexpr (BANDS && !GEN) {
        if (bands) {
            const bool lane_out = (lane >= C0_HXQ) && (lane < 64 - C0_HXQ);
            const int jo = x & 7;                                  // 0: left half of the cell, 4: right half
            const float wl0 = 1.0f - 0.125f * (float)jo;           // weight of column x towards the cell's left nodes; -1/8 per column
            float a0[4] = { 0.f, 0.f, 0.f, 0.f }, a1[4] = { 0.f, 0.f, 0.f, 0.f };
            const int Yc0 = y0 >> 3;
            // (a lane outside the exact output columns is never the left half of a written cell nor the partner of one: no
            // lane mask; rows outside the exact output are skipped as a whole, the test is wave-uniform)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int yr = wv * R + r, y = y0 + r;
                if (!(yr >= HY && yr < RH - HY && y >= 1 && y <= H - 2)) continue;
                const float v0 = (inner || x0ok) ? u[r].x : 0.f, v1 = (inner || x1ok) ? u[r].y : 0.f;
                const float v2 = (inner || x2ok) ? u[r].z : 0.f, v3 = (inner || x3ok) ? u[r].w : 0.f;
                const float r0 = __builtin_fmaf(wl0 - 0.375f, v3, __builtin_fmaf(wl0 - 0.25f, v2, __builtin_fmaf(wl0 - 0.125f, v1, wl0 * v0)));
                const float r1 = __builtin_fmaf(1.375f - wl0, v3, __builtin_fmaf(1.25f - wl0, v2, __builtin_fmaf(1.125f - wl0, v1, (1.0f - wl0) * v0)));
                const float tb = 0.125f * (float)(y & 7), tt = 1.0f - tb;
                if ((y >> 3) == Yc0) {
                    a0[0] = __builtin_fmaf(tt, r0, a0[0]); a0[1] = __builtin_fmaf(tt, r1, a0[1]);
                    a0[2] = __builtin_fmaf(tb, r0, a0[2]); a0[3] = __builtin_fmaf(tb, r1, a0[3]);
                } else {
                    a1[0] = __builtin_fmaf(tt, r0, a1[0]); a1[1] = __builtin_fmaf(tt, r1, a1[1]);
                    a1[2] = __builtin_fmaf(tb, r0, a1[2]); a1[3] = __builtin_fmaf(tb, r1, a1[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += wave_from_right(a0[k]); a1[k] += wave_from_right(a1[k]); }   // left half + right half of the cell
            const int Xc = x >> 3;
            if (jo == 0 && lane_out && x >= 0 && x < W && Xc < (P >> 3)) {
                float4 *o = bands + 2 * ((((size_t)c * nby + by) * NW + wv) * (size_t)(P >> 3) + Xc);
                o[0] = make_float4(a0[0], a0[1], a0[2], a0[3]);
                o[1] = make_float4(a1[0], a1[1], a1[2], a1[3]);
            }
        }
    }

    // ------------------------------------------------------------------ residual + restriction
#undef SC_C0_GS
    if (!FINAL) {
        constexpr int buf = (2 * T) & 1;       // the edges written after the last half-step
        const float4 up = (wv > 0) ? edge[buf][wv - 1][1][lane] : zero;
        const float4 dn = (wv < NW - 1) ? edge[buf][wv + 1][0][lane] : zero;
        const int I = x >> 1;                                  // coarse column of fine x (x even)
        // weights of the two fine points right of / below a coarse point: 1/2, 0 in general, the
        // interpolation-tail weights at the last coarse column / row (MGDim)
        const float wxa0 = (GEN && I == g.x.nc) ? g.x.tw1 : 0.5f, wxb0 = (GEN && I == g.x.nc) ? g.x.tw2 : 0.0f;
        const float wxa1 = (GEN && I + 1 == g.x.nc) ? g.x.tw1 : 0.5f, wxb1 = (GEN && I + 1 == g.x.nc) ? g.x.tw2 : 0.0f;
        float h0[R], h1[R];                    // horizontally filtered residual at coarse columns x and x+2
        auto residual_rows = [&](auto masked) {
        constexpr bool MK = decltype(masked)::value;      // inner waves: every point is interior
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + r;
            const bool yok = (y >= 1) && (y <= H - 2);
            (void)yok;
            const float4 a = (r == 0) ? up : u[r - 1];
            const float4 b = (r == R - 1) ? dn : u[r + 1];
            const float4 cur = u[r];
            float l = wave_from_left(cur.w), rr = wave_from_right(cur.x);
            // Level 0 (the only level whose residual has to be accurate: it decides what the solve converges
            // to) evaluates r = f - sum(neighbour - centre).  Each difference of two neighbouring values is
            // exact or very nearly so in float32 (Sterbenz: exact when they lie within a factor of two), so
            // the cancellation that the plain form  f - (sum(neighbours) - 4 centre)  suffers never happens.
            // The coarse levels solve for corrections; a float32 residual there perturbs the correction by a
            // relative 1e-7, which the next cycle's level-0 residual sees and removes.
#ifdef SC_RES_F64
            const double cn = (GEN && y == g.y.n) ? (double)g.y.cw_last : 1.0;
            const double dy = (GEN && y == g.y.n) ? (double)g.y.d_last : 2.0;
#define SC_C0_RES(L, R_, A, B, CC, K, CW, DX)                                                               \
    (float)((double)(HF ? c0_f_minus<K>(fh[HF ? r : 0], 0.0f) : GEN ? SC_C0_F(K) : -4.0f * SC_C0_F(K)) - ((((double)(CW) * (double)(L) + (double)(R_)) + (cn * (double)(A) + (double)(B))) - \
                            ((double)(DX) + dy) * (double)(CC)))
#else
            const float cn = (GEN && y == g.y.n) ? g.y.cw_last : 1.0f;
            const float dy = (GEN && y == g.y.n) ? g.y.d_last : 2.0f;
            (void)cn; (void)dy;
#define SC_C0_RES(L, R_, A, B, CC, K, CW, DX)                                                               \
    (GEN ? SC_C0_F(K) - ((((CW) * (L) + (R_)) + (cn * (A) + (B))) - ((DX) + dy) * (CC))                      \
     : HF ? c0_f_minus<K>(fh[HF ? r : 0], (((L) - (CC)) + ((R_) - (CC))) + (((A) - (CC)) + ((B) - (CC))))     \
          : __builtin_fmaf(SC_C0_F(K), -4.0f, -((((L) - (CC)) + ((R_) - (CC))) + (((A) - (CC)) + ((B) - (CC))))))
#endif
            float4 res;
            res.x = (!MK || (yok & x0ok)) ? SC_C0_RES(l, cur.y, a.x, b.x, cur.x, 0, cw0, dx0) : 0.f;
            res.y = (!MK || (yok & x1ok)) ? SC_C0_RES(cur.x, cur.z, a.y, b.y, cur.y, 1, cw1, dx1) : 0.f;
            res.z = (!MK || (yok & x2ok)) ? SC_C0_RES(cur.y, cur.w, a.z, b.z, cur.z, 2, cw2, dx2) : 0.f;
            res.w = (!MK || (yok & x3ok)) ? SC_C0_RES(cur.z, rr, a.w, b.w, cur.w, 3, cw3, dx3) : 0.f;
#undef SC_C0_RES
#undef SC_C0_F
            float rl = wave_from_left(res.w);
            if (GEN) {
                float rn = wave_from_right(res.x);
                h0[r] = ((0.5f * rl + res.x) + wxa0 * res.y) + wxb0 * res.z;
                h1[r] = ((0.5f * res.y + res.z) + wxa1 * res.w) + wxb1 * rn;
            } else {       // a weight of 1/2 is exact, so each fma rounds exactly where the product-then-sum form does
                h0[r] = __builtin_fmaf(res.y, 0.5f, __builtin_fmaf(rl, 0.5f, res.x));
                h1[r] = __builtin_fmaf(res.w, 0.5f, __builtin_fmaf(res.y, 0.5f, res.z));
            }
        }
        };
        if (inner) residual_rows(std::false_type{}); else residual_rows(std::true_type{});
        hedge[0][wv][lane] = make_float2(h0[R - 1], h1[R - 1]);
        if (GEN) hedge[1][wv][lane] = make_float2(h0[0], h1[0]);
        __syncthreads();
        const float2 hup = (wv > 0) ? hedge[0][wv - 1][lane] : make_float2(0.f, 0.f);
        const float2 hdn = (GEN && wv < NW - 1) ? hedge[1][wv + 1][lane] : make_float2(0.f, 0.f);
        const bool lane_out = (lane >= C0_HXQ) && (lane < 64 - C0_HXQ);
        float *__restrict__ fc = Fc.at(c);
        const float fx0 = (I == g.x.nc) ? 2.0f * g.x.inv_last : 1.0f;
        const float fx1 = (I + 1 == g.x.nc) ? 2.0f * g.x.inv_last : 1.0f;
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const int yr = wv * R + r, y = y0 + r;
            const int J = y >> 1;
            if (!lane_out || yr < HY || yr >= RH - HY || J < 1 || J > g.y.nc) continue;
            const float m0 = (r == 0) ? hup.x : h0[r - 1], m1 = (r == 0) ? hup.y : h1[r - 1];
            float v0, v1;
            if (GEN) {
                const float wya = (J == g.y.nc) ? g.y.tw1 : 0.5f, wyb = (J == g.y.nc) ? g.y.tw2 : 0.0f;
                const float p0 = (r + 2 < R) ? h0[(r + 2 < R) ? r + 2 : 0] : hdn.x;
                const float p1 = (r + 2 < R) ? h1[(r + 2 < R) ? r + 2 : 0] : hdn.y;
                v0 = ((0.5f * m0 + h0[r]) + wya * h0[r + 1]) + wyb * p0;
                v1 = ((0.5f * m1 + h1[r]) + wya * h1[r + 1]) + wyb * p1;
            } else {
                v0 = __builtin_fmaf(h0[r + 1], 0.5f, __builtin_fmaf(m0, 0.5f, h0[r]));
                v1 = __builtin_fmaf(h1[r + 1], 0.5f, __builtin_fmaf(m1, 0.5f, h1[r]));
            }
            const float fy = (J == g.y.nc) ? 2.0f * g.y.inv_last : 1.0f;
            if constexpr (L1H && !GEN) {
                __half *o = reinterpret_cast<__half *>(Fc.p) + (size_t)c * Fc.plane + (size_t)J * Fc.pitch + I;
                if (I >= 1 && I <= g.x.nc) o[0] = __float2half_rn(v0 * (fx0 * fy));
                if (I + 1 >= 1 && I + 1 <= g.x.nc) o[1] = __float2half_rn(v1 * (fx1 * fy));
            } else {
            float *o = fc + (size_t)J * Fc.pitch + I;
            if (I >= 1 && I <= g.x.nc) o[0] = v0 * (fx0 * fy);
            if (I + 1 >= 1 && I + 1 <= g.x.nc) o[1] = v1 * (fx1 * fy);
            }
        }
    }

    // ------------------------------------------------------------------ write back the exact inner tile
    if (lane < C0_HXQ || lane >= 64 - C0_HXQ || x >= P || x >= W) return;
    if constexpr (OUT) {
        // the result leaves as output values: u + node correction (the arithmetic of the post-process, sc_kernels.hip),
        // clamped to [0, 255], truncated; plane c of Uout's memory, rows of P bytes.  The splice kernel interleaves.
        uint8_t *__restrict__ q = reinterpret_cast<uint8_t *>(Uout.p) + (size_t)c * Uout.plane;
        // The lane's four pixels lie in ONE 8-column cell and its R = 8 rows in at most two cell rows: six node values serve the
        // whole band (round 4; four loads per ROW before: 32 dependent-latency loads at the very end of the launch).  Requested
        // together, clamped into the node grid; lm_add4_nodes is lm_add4's arithmetic, operation for operation.
        const bool corr = lm.CN && x <= W - 2;   // the lane that holds only the ring column x = W - 1 would read node (x >> 3) + 1 == nx, one past the row (its byte is never spliced)
        const int Yc0 = max(y0, 0) >> 3;
        float nd[3][2] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
        if (corr) {
            const float *__restrict__ pn = lm.CN + (size_t)c * lm.ny * lm.npitch + (x >> 3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float *row = pn + (size_t)min(Yc0 + k, lm.ny - 1) * lm.npitch;
                nd[k][0] = row[0]; nd[k][1] = row[1];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yr = wv * R + r, y = y0 + r;
            if (!(yr >= HY && yr < RH - HY && y >= 1 && y <= H - 2)) continue;
            float4 v = u[r];
            if (corr) {
                const bool second = (y >> 3) != Yc0;       // wave-uniform: the band's rows lie in cell rows Yc0 and Yc0 + 1
                lm_add4_nodes(second ? nd[1][0] : nd[0][0], second ? nd[1][1] : nd[0][1], second ? nd[2][0] : nd[1][0], second ? nd[2][1] : nd[1][1], x, y, v);
            }
            *reinterpret_cast<unsigned *>(q + (size_t)y * P + x) = lm_byte(v.x) | (lm_byte(v.y) << 8) | (lm_byte(v.z) << 16) | (lm_byte(v.w) << 24);
        }
        return;
    }
    if constexpr (GEN && L1H) {      // the level-1 correction leaves as float16: level 0's prolongation reads it that way
        __half *__restrict__ outh = reinterpret_cast<__half *>(Uout.p) + (size_t)c * Uout.plane;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yr = wv * R + r, y = y0 + r;
            if (!(yr >= HY && yr < RH - HY && y >= 0 && y < H)) continue;
            const __half2 a = __floats2half2_rn(u[r].x, u[r].y), b2 = __floats2half2_rn(u[r].z, u[r].w);
            uint2 pk; pk.x = *reinterpret_cast<const unsigned *>(&a); pk.y = *reinterpret_cast<const unsigned *>(&b2);
            *reinterpret_cast<uint2 *>(outh + (size_t)y * P + x) = pk;
        }
        return;
    }
    if constexpr (UQO) {
        uint16_t *__restrict__ outq = reinterpret_cast<uint16_t *>(Uout.p) + (size_t)c * Uout.plane;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;          // per column: pad columns right of the ring hold nothing the solve uses
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yr = wv * R + r, y = y0 + r;
            if (!(yr >= HY && yr < RH - HY && y >= 0 && y < H)) continue;
            uint2 pk;
            pk.x = c0_q16_checked(u[r].x, s0) | (c0_q16_checked(u[r].y, s1) << 16);
            pk.y = c0_q16_checked(u[r].z, s2) | (c0_q16_checked(u[r].w, s3) << 16);
            *reinterpret_cast<uint2 *>(outq + (size_t)y * P + x) = pk;
        }
        const int seen = s0 | (x + 1 < W ? s1 : 0) | (x + 2 < W ? s2 : 0) | (x + 3 < W ? s3 : 0);
        if ((seen & ~0xffff) && sat.p) { *sat.p = sat.gen; if (sat.host) *sat.host = sat.gen; }       // rare; every writer stores the same word
        return;
    }
    float *__restrict__ out = Uout.at(c);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yr = wv * R + r, y = y0 + r;
        if (yr >= HY && yr < RH - HY && y >= 0 && y < H) *reinterpret_cast<float4 *>(out + (size_t)y * P + x) = u[r];
    }
}

constexpr int C0_NW = 8, C0_R = 8;

// the level-0 forms a size class can meet (the default fast path and its continuation on float fields): first launch 134 | 512 / 134,
// catch-up 130, full cycles 146 | 768 / 146, full cycles leaving cell shares 210 | 256 / 210, final cycle as a field 154, as bytes 186
constexpr bool c0_rag_form(int TAG)
{
    return TAG == (134 | 512) || TAG == 134 || TAG == 130 || TAG == (146 | 768) || TAG == 146 || TAG == (210 | 256) || TAG == 210 ||
           TAG == 154 || TAG == 186;
}

// rag != nullptr: the launch serves a size class (k_cycle0 TAG bit 10); -1 where that form is not instantiated
template <int T, bool PRO, int TAG = 0, int NW = C0_NW>
static int launch_c0(Field Uin, Field Uout, Field F, Field Fc, Field E, const MGGeom &g, float *partial, hipStream_t s,
                     const ComposeArgs &comp = ComposeArgs(), float4 *bands = nullptr, const LmNodes &lm = LmNodes(), const AbortFlag &sat = AbortFlag(),
                     const RagMember *rag = nullptr)
{
    constexpr int RH = NW * C0_R, HY = 2 * T + 2;
    const int blocks = ((Uin.W + (256 - 2 * C0_HX) - 1) / (256 - 2 * C0_HX)) * ((Uin.H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * Uin.C;
    if (rag) {
        if constexpr (c0_rag_form(TAG) && NW == C0_NW)
            hipLaunchKernelGGL((k_cycle0<T, NW, C0_R, PRO, false, false, TAG | 1024>), dim3(blocks), dim3(NW * 64), 0, s, Uin, Uout, F, Fc, E,
                               g, partial, comp, bands, lm, sat, rag, 0);
        else return -1;
        return blocks;
    }
    hipLaunchKernelGGL((k_cycle0<T, NW, C0_R, PRO, false, false, TAG>), dim3(blocks), dim3(NW * 64), 0, s, Uin, Uout, F, Fc, E,
                       g, partial, comp, bands, lm, sat, (const RagMember *)nullptr, 0);
    return blocks;
}

// (Round 4, measured and not kept: sixteen-wave workgroups -- 128-row windows, one per CU -- for a single clone's two-sweep level-0
// launches, 486 workgroups on 256 slots instead of 1080 on 512: 30.6 us against 29.0.  The launch is not paced by rounds of
// workgroup slots; level 1's launch is, see launch_cycle_coarse.)

// Level-0 launch whose prolongation source is composed on the fly: U1 = level-1 correction after its pre-smoothing (level 1
// has no post-smoothing and no prolongation launch of its own), E2 = finished level-2 correction, g1 = level-1 geometry.
// sweeps = post + pre (4) or, final_cycle, post (2).  Returns the number of partial maxima, -1 if not instantiated.
// With a float16 right-hand side (f_half: the fast path) level 1's fields are float16 as well (TAG bit 7; sc_multigrid.cpp decides
// with the same rule: mg_level1_half).  u_q16 (float16 level 1, full cycles only): Uin holds 16-bit fixed point (c0_load_q16);
// bit 1 (value 2) set: so will Uout, clear: Uout leaves as float (the launch before the judged cycle).
int launch_cycle0_composed(Field Uin, Field Uout, Field F, Field Fc, Field U1, const MGGeom &g, int sweeps, float *partial,
                           hipStream_t s, bool tag, bool f_half, bool final_cycle, Field E2, const MGGeom &g1, float4 *bands, bool l1_half,
                           int u_q16, AbortFlag sat, const RagMember *rag)
{
    ComposeArgs ca;
    ca.E2 = E2; ca.g1 = g1;
    if (rag) {      // a size class: the default fast path's forms only (float16 right-hand side and level 1)
        if (!l1_half || !f_half || tag) return -1;
        if (final_cycle) return (sweeps == 2 && !u_q16) ? launch_c0<2, true, 154>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands, LmNodes(), AbortFlag(), rag) : -1;
        if (sweeps != 4) return -1;
        if (u_q16 == 3) return bands ? -1 : launch_c0<4, true, 146 | 768>(Uin, Uout, F, Fc, U1, g, partial, s, ca, nullptr, LmNodes(), sat, rag);
        if (u_q16 == 1) return bands ? launch_c0<4, true, 210 | 256>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands, LmNodes(), AbortFlag(), rag) : -1;
        if (u_q16) return -1;
        return bands ? launch_c0<4, true, 210>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands, LmNodes(), AbortFlag(), rag)
                     : launch_c0<4, true, 146>(Uin, Uout, F, Fc, U1, g, partial, s, ca, nullptr, LmNodes(), AbortFlag(), rag);
    }
    if (u_q16) {
        if (!l1_half || !f_half || final_cycle || sweeps != 4) return -1;
        if (!(u_q16 & 2)) {
            if (tag) return -1;
            return bands ? launch_c0<4, true, 210 | 256>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands) : launch_c0<4, true, 146 | 256>(Uin, Uout, F, Fc, U1, g, partial, s, ca);
        }
        if (bands && !tag) return launch_c0<4, true, 210 | 768>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands, LmNodes(), sat);
        return tag ? launch_c0<4, true, 147 | 768>(Uin, Uout, F, Fc, U1, g, partial, s, ca, nullptr, LmNodes(), sat)
                   : launch_c0<4, true, 146 | 768>(Uin, Uout, F, Fc, U1, g, partial, s, ca, nullptr, LmNodes(), sat);
    }
    if (l1_half != f_half) {           // instantiated pairs: float16 RHS with float16 level 1, float RHS with float level 1 -- and, for
        if (l1_half) return -1;        // a level 1 that does fewer than four sweeps (mg_level1_sweeps), float16 RHS with float level 1
        if (final_cycle) return sweeps == 2 ? launch_c0<2, true, 26>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands) : -1;
        if (sweeps != 4) return -1;
        if (bands && !tag) return launch_c0<4, true, 82>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands);
        return tag ? launch_c0<4, true, 19>(Uin, Uout, F, Fc, U1, g, partial, s, ca) : launch_c0<4, true, 18>(Uin, Uout, F, Fc, U1, g, partial, s, ca);
    }
    if (final_cycle) {
        if (sweeps != 2) return -1;
        return f_half ? launch_c0<2, true, 154>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands) : launch_c0<2, true, 24>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands);
    }
    if (sweeps != 4) return -1;
    if (bands && !tag) return f_half ? launch_c0<4, true, 210>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands) : launch_c0<4, true, 80>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands);
    if (tag) return f_half ? launch_c0<4, true, 147>(Uin, Uout, F, Fc, U1, g, partial, s, ca) : launch_c0<4, true, 17>(Uin, Uout, F, Fc, U1, g, partial, s, ca);
    return f_half ? launch_c0<4, true, 146>(Uin, Uout, F, Fc, U1, g, partial, s, ca) : launch_c0<4, true, 16>(Uin, Uout, F, Fc, U1, g, partial, s, ca);
}

// sweeps = T red-black GS sweeps; prolong: add P*E first and write per-block max|P*E| to `partial`; f_half / u_half: F /
// Uin hold float16 values (same element layout).  Returns the number of partial maxima written (0 without prolong), or -1
// for an unsupported depth.
int launch_cycle0(Field Uin, Field Uout, Field F, Field Fc, Field E, const MGGeom &g, int sweeps, bool prolong,
                  float *partial, hipStream_t s, bool tag, bool f_half, bool u_half, bool final_cycle, float4 *bands, bool l1_half, bool q16_out, AbortFlag sat,
                  const RagMember *rag)
{
    if (q16_out && !(l1_half && u_half)) return -1;      // the first launch of a clone on the fast path only
    if (rag && !l1_half) return -1;                      // a size class: the fast path's forms only
    if (l1_half) {     // level 1 keeps float16 fields (the composed schedule): the launches without a prolongation write its right-hand side
        if (prolong || final_cycle || !f_half || tag || sweeps != 2) return -1;
        if (q16_out) launch_c0<2, false, 134 | 512>(Uin, Uout, F, Fc, E, g, partial, s, ComposeArgs(), nullptr, LmNodes(), sat, rag);
        else if (u_half) launch_c0<2, false, 134>(Uin, Uout, F, Fc, E, g, partial, s, ComposeArgs(), nullptr, LmNodes(), AbortFlag(), rag);
        else launch_c0<2, false, 130>(Uin, Uout, F, Fc, E, g, partial, s, ComposeArgs(), nullptr, LmNodes(), AbortFlag(), rag);
        return 0;
    }
    if (final_cycle) {   // prolongation + `sweeps` post-smoothing sweeps, nothing restricted
        if (!prolong || u_half) return -1;
        const ComposeArgs nc = ComposeArgs();
        switch (sweeps) {
        case 1: return f_half ? launch_c0<1, true, 10>(Uin, Uout, F, Fc, E, g, partial, s, nc, bands) : launch_c0<1, true, 8>(Uin, Uout, F, Fc, E, g, partial, s, nc, bands);
        case 2: return f_half ? launch_c0<2, true, 10>(Uin, Uout, F, Fc, E, g, partial, s, nc, bands) : launch_c0<2, true, 8>(Uin, Uout, F, Fc, E, g, partial, s, nc, bands);
        default: return -1;
        }
    }
    if (u_half) {      // first launch of a clone on the float16 fields the pre-process wrote
        if (prolong || !f_half) return -1;
        switch (sweeps) {
        case 1: launch_c0<1, false, 6>(Uin, Uout, F, Fc, E, g, partial, s); return 0;
        case 2: launch_c0<2, false, 6>(Uin, Uout, F, Fc, E, g, partial, s); return 0;
        default: return -1;
        }
    }
#define SC_C0(T_, PRO_) (f_half ? (tag ? launch_c0<T_, PRO_, 3>(Uin, Uout, F, Fc, E, g, partial, s)      \
                                       : launch_c0<T_, PRO_, 2>(Uin, Uout, F, Fc, E, g, partial, s))      \
                                : (tag ? launch_c0<T_, PRO_, 1>(Uin, Uout, F, Fc, E, g, partial, s)      \
                                       : launch_c0<T_, PRO_, 0>(Uin, Uout, F, Fc, E, g, partial, s)))
    if (prolong && bands && sweeps == 4 && !tag)
        return f_half ? launch_c0<4, true, 66>(Uin, Uout, F, Fc, E, g, partial, s, ComposeArgs(), bands) : launch_c0<4, true, 64>(Uin, Uout, F, Fc, E, g, partial, s, ComposeArgs(), bands);
    if (prolong) {
        switch (sweeps) {
        case 2: return SC_C0(2, true);
        case 3: return SC_C0(3, true);
        case 4: return SC_C0(4, true);
        default: return -1;
        }
    }
    switch (sweeps) {
    case 1: SC_C0(1, false); return 0;
    case 2: SC_C0(2, false); return 0;
    default: return -1;
    }
#undef SC_C0
}

// The last cycle of a clone with its output leaving as bytes (TAG bit 5): prolongation (composed: E = U1 with E2 / g1, else
// E = the finished level-1 correction) + two post-smoothing sweeps; Q (a field's memory: plane c at Q.p + c Q.plane BYTES,
// rows of Q.pitch bytes) receives the output values, lm the node correction to add (CN == nullptr: none).
int launch_cycle0_out(Field Uin, Field Q, Field F, Field Fc, Field E, const MGGeom &g, float *partial, hipStream_t s, bool f_half,
                      bool composed, Field E2, const MGGeom &g1, const LmNodes &lm, bool l1_half, const RagMember *rag)
{
    ComposeArgs ca;
    if (l1_half && !(composed && f_half)) return -1;
    if (rag) {
        if (!(composed && f_half && l1_half)) return -1;
        ca.E2 = E2; ca.g1 = g1;
        return launch_c0<2, true, 186>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm, AbortFlag(), rag);
    }
    if (composed && f_half && !l1_half) {
        ca.E2 = E2; ca.g1 = g1;
        return launch_c0<2, true, 58>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm);
    }
    if (composed) {
        ca.E2 = E2; ca.g1 = g1;
        return f_half ? launch_c0<2, true, 186>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm) : launch_c0<2, true, 56>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm);
    }
    return f_half ? launch_c0<2, true, 42>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm) : launch_c0<2, true, 40>(Uin, Q, F, Fc, E, g, partial, s, ca, nullptr, lm);
}

// The other three level-0 launches of a fast-path solve under a second symbol (TAG bit 0), for isolated timing (sc_hip_time_cycle0_form):
// form 1 = the full cycle before the judged one (16-bit field in, float out, leaves the correction's cell shares: in-step symbol
// ..., 466>), 2 = the judged cycle writing output bytes (..., 186>), 3 = the first launch of a solve (float16 initial field in,
// 16-bit field out, no prolongation: ..., 646>).  Form 0 (..., 914> -> 915>) is launch_cycle0_composed(tag = true).
int launch_cycle0_twin(int form, Field Uin, Field Uout, Field F, Field Fc, Field U1, const MGGeom &g, float *partial, hipStream_t s,
                       Field E2, const MGGeom &g1, float4 *bands, const LmNodes &lm)
{
    ComposeArgs ca;
    ca.E2 = E2; ca.g1 = g1;
    switch (form) {
    case 1: return launch_c0<4, true, 210 | 256 | 1>(Uin, Uout, F, Fc, U1, g, partial, s, ca, bands);
    case 2: return launch_c0<2, true, 186 | 1>(Uin, Uout, F, Fc, U1, g, partial, s, ca, nullptr, lm);
    case 3: launch_c0<2, false, 134 | 512 | 1>(Uin, Uout, F, Fc, Field(), g, partial, s); return 0;
    default: return -1;
    }
}

// tiling in y of a level-0 launch with `sweeps` sweeps: nby tile rows, tile row b processes field rows [b step - hy, + 64) in
// eight 8-row bands (one per wave) and owns the output rows [b step, (b + 1) step)
void cycle0_row_geometry(int H, int sweeps, int &nby, int &step, int &hy)
{
    const int RH = C0_NW * C0_R;
    hy = 2 * sweeps + 2; step = RH - 2 * hy; nby = (H + step - 1) / step;
}

int cycle0_blocks(int W, int H, int C, int sweeps)
{
    const int RH = C0_NW * C0_R, HY = 2 * sweeps + 2;
    return ((W + (256 - 2 * C0_HX) - 1) / (256 - 2 * C0_HX)) * ((H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * C;
}

// Coarse levels (l >= 1): pre-smoothing from a zero correction + residual + restriction in one
// launch.  Uout receives the smoothed correction, Fc the next level's RHS.
// rag / lev: a size class (k_cycle0 TAG bit 10): level `lev` of every member's own hierarchy; instantiated for the depths the default
// schedule uses -- four sweeps on float16 fields (level 1) and two on float ones (the levels below it)
template <int T, int R, int TAG = 0, int NW = C0_NW>
static bool launch_cn(Field Uout, Field F, Field Fc, const MGGeom &g, hipStream_t s, const RagMember *rag = nullptr, int lev = 0)
{
    constexpr int RH = NW * R, HY = 2 * T + 2;
    Field none{};
    const int blocks = ((F.W + (256 - 2 * C0_HX) - 1) / (256 - 2 * C0_HX)) * ((F.H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * F.C;
    if (rag) {
        if constexpr ((T == 4 && TAG == 128 && R != 8) || (T == 2 && TAG == 0 && NW == C0_NW))
            hipLaunchKernelGGL((k_cycle0<T, NW, R, false, true, true, TAG | 1024>), dim3(blocks), dim3(NW * 64), 0, s, F, Uout, F, Fc, none, g,
                               (float *)nullptr, ComposeArgs(), (float4 *)nullptr, LmNodes(), AbortFlag(), rag, lev);
        else return false;
        return true;
    }
    hipLaunchKernelGGL((k_cycle0<T, NW, R, false, true, true, TAG>), dim3(blocks), dim3(NW * 64), 0, s, F /*unused Uin: geometry only*/,
                       Uout, F, Fc, none, g, (float *)nullptr, ComposeArgs(), (float4 *)nullptr, LmNodes(), AbortFlag(), (const RagMember *)nullptr, 0);
    return true;
}

// workgroups of a coarse-level launch with NW waves of R rows each at depth T
static int cn_blocks(const Field &F, int T, int R, int NW)
{
    const int RH = NW * R, HY = 2 * T + 2;
    return ((F.W + (256 - 2 * C0_HX) - 1) / (256 - 2 * C0_HX)) * ((F.H + (RH - 2 * HY) - 1) / (RH - 2 * HY)) * F.C;
}

// half_io: the level's own right-hand side and the correction it writes are float16 (level 1 of the composed schedule, 4 sweeps)
bool launch_cycle_coarse(Field Uout, Field F, Field Fc, const MGGeom &g, int sweeps, hipStream_t s, bool half_io, const RagMember *rag, int lev)
{
    if (rag) {          // a size class: the same choices from the class's dimensions
        if (half_io) {
            if (sweeps != 4) return false;
            const int R4 = tb_gen_rows_deep(F.W, F.H, F.C, C0_HX, 2 * sweeps + 2);
            const long b8 = cn_blocks(F, 4, 6, 8), b16 = cn_blocks(F, 4, 6, 16);
            if (R4 == 6 && ((b8 > 512 && b16 <= 256) || (F.C > 3 && 2 * b16 < b8))) return launch_cn<4, 6, 128, 16>(Uout, F, Fc, g, s, rag, lev);
            return R4 == 6 ? launch_cn<4, 6, 128>(Uout, F, Fc, g, s, rag, lev) : launch_cn<4, 4, 128>(Uout, F, Fc, g, s, rag, lev);
        }
        if (sweeps != 2) return false;
        const int R = tb_gen_rows(F.W, F.H, F.C, C0_HX, 2 * sweeps + 2);
        return R == 8 ? launch_cn<2, 8>(Uout, F, Fc, g, s, rag, lev) : R == 6 ? launch_cn<2, 6>(Uout, F, Fc, g, s, rag, lev) : launch_cn<2, 4>(Uout, F, Fc, g, s, rag, lev);
    }
    if (half_io) {
        if (sweeps != 4) return false;
        const int R4 = tb_gen_rows_deep(F.W, F.H, F.C, C0_HX, 2 * sweeps + 2);
        // Round 4: a single clone's level 1 at 2048^2 is 555 eight-wave workgroups on 512 slots (two per CU): 43 of them make a second
        // round and the launch takes 18 us for 10 us of work.  Sixteen-wave workgroups (96-row windows, 76 of them exact instead of
        // 28 of 48; one per CU) need 210: one round.  Taken whenever it turns more than one round of the 8-wave form into one.
        // ... and for a GROUP of clones (tens of rounds either way) whenever the 16-wave tiling needs fewer waves in total: 76 of 96
        // rows exact instead of 28 of 48 (+0.7 % on the bench step, tools/ab_step.py).
        const long b8 = cn_blocks(F, 4, 6, 8), b16 = cn_blocks(F, 4, 6, 16);
        if (R4 == 6 && ((b8 > 512 && b16 <= 256) || (F.C > 3 && 2 * b16 < b8))) { launch_cn<4, 6, 128, 16>(Uout, F, Fc, g, s); return true; }
        R4 == 6 ? launch_cn<4, 6, 128>(Uout, F, Fc, g, s) : launch_cn<4, 4, 128>(Uout, F, Fc, g, s);
        return true;
    }
    if (sweeps < 1 || sweeps > 4) return false;
    const int R = sweeps >= 3 ? tb_gen_rows_deep(F.W, F.H, F.C, C0_HX, 2 * sweeps + 2) : tb_gen_rows(F.W, F.H, F.C, C0_HX, 2 * sweeps + 2);
    if (sweeps == 1) { R == 8 ? launch_cn<1, 8>(Uout, F, Fc, g, s) : R == 6 ? launch_cn<1, 6>(Uout, F, Fc, g, s) : launch_cn<1, 4>(Uout, F, Fc, g, s); }
    else if (sweeps == 2) { R == 8 ? launch_cn<2, 8>(Uout, F, Fc, g, s) : R == 6 ? launch_cn<2, 6>(Uout, F, Fc, g, s) : launch_cn<2, 4>(Uout, F, Fc, g, s); }
    else {
        // deeper pre-smoothing (a level that gets no post-smoothing: 3 or 4 sweeps): 4- or 6-row bands (tb_gen_rows_deep);
        // 8-row bands spill with the general coefficients (23 us against 20.6 us for level 1 of a 2048^2 ROI)
        if (sweeps == 3) { R == 6 ? launch_cn<3, 6>(Uout, F, Fc, g, s) : launch_cn<3, 4>(Uout, F, Fc, g, s); }
        else             { R == 6 ? launch_cn<4, 6>(Uout, F, Fc, g, s) : launch_cn<4, 4>(Uout, F, Fc, g, s); }
    }
    return true;
}

} // namespace sc
