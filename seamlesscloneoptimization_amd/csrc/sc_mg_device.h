// sc_mg_device.h -- device helpers shared by the multigrid kernels: 1-D restriction weights and
// 1-D interpolation stencils for a level pair described by MGDim (sc_common.h).
#pragma once
#include "sc_common.h"

namespace sc {

// Coarse point I gathers fine points 2I-1, 2I, 2I+1 with weights 1/2, 1, 1/2; the last coarse
// point takes the (up to two) tail points with their interpolation weights instead.  `inv` is
// the reciprocal row sum (the restriction is the row-normalised transpose of the interpolation).
__device__ __forceinline__ void restrict_weights(const MGDim &d, int I, float w[4], float &inv)
{
    w[0] = 0.5f; w[1] = 1.0f;
    if (I < d.nc) { w[2] = 0.5f; w[3] = 0.0f; inv = 0.5f; }
    else          { w[2] = d.tw1; w[3] = d.tw2; inv = d.inv_last; }
}


// Interpolation stencil of fine point i: coarse indices I0, I1 and weights w0, w1.
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


// ---------------------------------------------------------------------------------------------------------------
// Bilinear prolongation for the register-blocked kernels (lane = 4 fine columns x R fine rows, x % 4 == 0, y0 even).
//
// The lane interpolates from coarse columns x/2 .. x/2+2 and rows y0/2 .. y0/2+R/2.  The last interval of a level is
// irregular (MGDim): the one or two fine points beyond the last coarse point nc interpolate between E[nc] and the
// boundary value 0 with weights tw1 / tw2.  Both equal the REGULAR formula applied to a ghost value
//     E[nc + 1] := (2 tw1 - 1) E[nc]
// (tw2 = 2 tw1 - 1: the tail points lie on one straight line to the boundary), so one branch-free code path serves
// every lane: the ghost column / row is patched in where the lane's window contains index nc + 1.  (The former
// "general" path for such lanes made every workgroup on the image border a straggler.)
//
// COMP: the level interpolated FROM (level l+1) ran without post-smoothing and without a prolongation launch of its own;
// its finished correction is  E = U(l+1) + P E(l+2)  and is formed here on the fly from the rows of level l+2 under the
// window, with the same ghost rule one level up (ComposeArgs: E2 = finished level l+2 correction, g1 = geometry of l+1).
// LEFT: also fetch column x/2 - 1 -- needed only where a level can have TWO tail points (x/2 itself is then the ghost
// column for the lane that starts on the last fine point); level 0 has at most one.
// ---------------------------------------------------------------------------------------------------------------
template <int R, bool COMP, bool LEFT>
struct ProlongWindow {
    static constexpr int NE = R / 2 + 2, NQ = R / 4 + 3;
    float2 ab[NE];                 // coarse columns x/2, x/2+1 of rows y0/2 - 1 .. y0/2 + R/2 (one extra row above: ghost source)
    float cc[NE];                  // column x/2 + 2
    float lf[LEFT ? NE : 1];       // column x/2 - 1
    float2 q[COMP ? NQ : 1];       // level l+2: columns x/4, x/4+1 of the rows under the window
    float ql[COMP ? NQ : 1];       // level l+2: column x/4 - 1
};

// issue the loads (clamped indices: what a clamped index reads is a zero ring / pad value or is never used)
template <int R, bool COMP, bool LEFT>
__device__ __forceinline__ void prolong_load(ProlongWindow<R, COMP, LEFT> &w, const Field &E, const ComposeArgs &comp, int c, int x, int y0)
{
    using PW = ProlongWindow<R, COMP, LEFT>;
    const float *__restrict__ e = E.at(c);
    const int cx = min(max(x >> 1, 0), E.pitch - 4), J = (y0 >> 1) - 1;
#pragma unroll
    for (int j = 0; j < PW::NE; ++j) {
        const float *er = e + (size_t)min(max(J + j, 0), E.H - 1) * E.pitch + cx;
        w.ab[j] = *reinterpret_cast<const float2 *>(er);
        w.cc[j] = er[2];
        if (LEFT) w.lf[j] = er[cx > 0 ? -1 : 0];
    }
    if (COMP) {
        const float *__restrict__ e2 = comp.E2.at(c);
        const int qx = min(max(x >> 2, 0), comp.E2.pitch - 2), Q = ((y0 >> 1) - 1) >> 1;
#pragma unroll
        for (int k = 0; k < PW::NQ; ++k) {
            const float *er = e2 + (size_t)min(max(Q + k, 0), comp.E2.H - 1) * comp.E2.pitch + qx;
            w.q[k] = make_float2(er[0], er[1]);
            w.ql[k] = er[qx > 0 ? -1 : 0];
        }
    }
}

// u += P E on the lane's interior points; returns max |P E| over them.  W, H: fine field size (ring included).
template <int R, bool COMP, bool LEFT>
__device__ __forceinline__ float prolong_apply(const ProlongWindow<R, COMP, LEFT> &w, const MGGeom &g, const ComposeArgs &comp,
                                               int x, int y0, int W, int H, float4 (&u)[R])
{
    using PW = ProlongWindow<R, COMP, LEFT>;
    constexpr int NE = PW::NE, NQ = PW::NQ;
    float m = 0.f;
    if (x >= 0 && x <= W - 2 && y0 + R - 1 >= 1 && y0 <= H - 2) {        // the lane owns at least one interior point
    // the same expressions as in the calling kernels, so that the masks are shared after inlining (SGPR pressure)
    const bool x0ok = (x + 0 >= 1) && (x + 0 <= W - 2), x1ok = (x + 1 >= 1) && (x + 1 <= W - 2);
    const bool x2ok = (x + 2 >= 1) && (x + 2 <= W - 2), x3ok = (x + 3 >= 1) && (x + 3 <= W - 2);
    const int ncx = g.x.nc, ncy = g.y.nc;
    const float gx = 2.0f * g.x.tw1 - 1.0f, gy = 2.0f * g.y.tw1 - 1.0f;
    const int c0 = x >> 1, Jb = (y0 >> 1) - 1;                      // first coarse column / first loaded coarse row
    float p0[COMP ? NE : 1], p1[COMP ? NE : 1], p2[COMP ? NE : 1], pl[(COMP && LEFT) ? NE : 1];   // P E(l+2) at the window's columns, per row
    if (COMP) {
        const int n2x = comp.g1.x.nc, n2y = comp.g1.y.nc, q0 = x >> 2, Qb = Jb >> 1;
        const float g1x = 2.0f * comp.g1.x.tw1 - 1.0f, g1y = 2.0f * comp.g1.y.tw1 - 1.0f;
        float h0[NQ], h1[NQ], h2[NQ], hl[LEFT ? NQ : 1];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            float a2 = (q0 <= n2x) ? w.q[k].x : 0.f, b2 = (q0 + 1 <= n2x) ? w.q[k].y : 0.f;
            const float z2 = w.ql[k];                               // read only where q0 - 1 <= n2x
            if (q0 == n2x) b2 = g1x * a2;                           // ghost column of level l+2
            if (q0 == n2x + 1) a2 = g1x * z2;
            h0[k] = a2; h1[k] = 0.5f * a2 + 0.5f * b2; h2[k] = b2;
            if (LEFT) hl[k] = 0.5f * z2 + 0.5f * a2;                // column x/2 - 1 (odd: between x/4 - 1 and x/4)
        }
#pragma unroll
        for (int k = 1; k < NQ; ++k)
            if (Qb + k == n2y + 1) {                                // ghost row
                h0[k] = g1y * h0[k - 1]; h1[k] = g1y * h1[k - 1]; h2[k] = g1y * h2[k - 1];
                if (LEFT) hl[k] = g1y * hl[k - 1];
            }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int Jr = Jb + j, k = (Jr >> 1) - Qb;              // row of level l+1 and the level l+2 row at / above it
            if (Jr & 1) {
                p0[j] = 0.5f * h0[k] + 0.5f * h0[k + 1]; p1[j] = 0.5f * h1[k] + 0.5f * h1[k + 1]; p2[j] = 0.5f * h2[k] + 0.5f * h2[k + 1];
                if (LEFT) pl[j] = 0.5f * hl[k] + 0.5f * hl[k + 1];
            } else {
                p0[j] = h0[k]; p1[j] = h1[k]; p2[j] = h2[k];
                if (LEFT) pl[j] = hl[k];
            }
        }
    }
    float4 row[NE];      // coarse row j of the window, interpolated in x to the lane's four fine columns
    float3 prev = make_float3(0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int Jr = Jb + j;
        float ea = w.ab[j].x, eb = w.ab[j].y, ec = w.cc[j], el = LEFT ? w.lf[j] : 0.f;
        if (COMP) {      // correction of level l+1 = its pre-smoothed U + interpolated correction of level l+2, zero outside the level
            const bool rin = (Jr >= 1) && (Jr <= ncy);
            ea = (rin && c0 >= 1 && c0 <= ncx) ? ea + p0[j] : 0.f;
            eb = (rin && c0 + 1 <= ncx) ? eb + p1[j] : 0.f;
            ec = (rin && c0 + 2 <= ncx) ? ec + p2[j] : 0.f;
            if (LEFT) el = (rin && c0 - 1 >= 1 && c0 - 1 <= ncx) ? el + pl[j] : 0.f;
        }
        if (LEFT && c0 == ncx + 1) ea = gx * el;                    // ghost column of this level
        if (c0 == ncx) eb = gx * ea;
        if (c0 + 1 == ncx) ec = gx * eb;
        if (Jr == ncy + 1) { ea = gy * prev.x; eb = gy * prev.y; ec = gy * prev.z; }   // ghost row
        prev = make_float3(ea, eb, ec);
        row[j] = make_float4(ea, 0.5f * ea + 0.5f * eb, eb, 0.5f * eb + 0.5f * ec);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        float4 cr = row[r / 2 + 1];
        if (r & 1) {
            const float4 nx = row[r / 2 + 2];
            cr = make_float4(0.5f * cr.x + 0.5f * nx.x, 0.5f * cr.y + 0.5f * nx.y, 0.5f * cr.z + 0.5f * nx.z, 0.5f * cr.w + 0.5f * nx.w);
        }
        if (y < 1 || y > H - 2) continue;
        if (x0ok) { u[r].x = u[r].x + cr.x; m = fmaxf(m, fabsf(cr.x)); }
        if (x1ok) { u[r].y = u[r].y + cr.y; m = fmaxf(m, fabsf(cr.y)); }
        if (x2ok) { u[r].z = u[r].z + cr.z; m = fmaxf(m, fabsf(cr.z)); }
        if (x3ok) { u[r].w = u[r].w + cr.w; m = fmaxf(m, fabsf(cr.w)); }
    }
    }
    return m;
}

} // namespace sc
