// sc_multigrid.cpp -- geometric multigrid V-cycle for the ROI Poisson system (SURVEY.md
// section 8 row f1): red-black GS smoothing, residual in double, normalised-transpose
// restriction, bilinear prolongation.  Arbitrary ROI sizes coarsen by letting the LAST grid
// interval of each level differ from the others (MGDim), so the Dirichlet ring never moves.
// Level 0 runs the exact 5-point kernels of the sweep solvers; coarser levels the general
// ones.  Converges ~20x per V(2,2) cycle at every size tried (tools/mg_proto2.py).
#include "sc_instance.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace sc {

static void coarsen_1d(int n, double a, int &nc, double &ac)
{
    if (n % 2 == 1) { nc = (n - 1) / 2; ac = (1.0 + a) / 2.0; }       // boundary stays (1+a)/2 coarse cells away
    else if (a >= 1.0) { nc = n / 2; ac = a / 2.0; }                   // keep the last point
    else { nc = n / 2 - 1; ac = 1.0 + a / 2.0; }                       // drop it: gap would fall below 1/2
}

static MGDim make_dim(int n, double a, int nc)
{
    MGDim d;
    d.n = n; d.nc = nc; d.alpha = (float)a;
    d.cw_last = (float)(2.0 / (1.0 + a));
    d.d_last = (float)(2.0 / a);
    const int tail = n - 2 * nc;          // 0, 1 or 2 fine points beyond the last coarse point
    const double D = n + a - 2.0 * nc;    // their distance budget to the boundary
    d.tw1 = tail >= 1 ? (float)(1.0 - 1.0 / D) : 0.f;
    d.tw2 = tail >= 2 ? (float)(1.0 - 2.0 / D) : 0.f;
    d.inv_last = (float)(1.0 / (1.5 + (double)d.tw1 + (double)d.tw2));
    return d;
}

static Field level_field(void *p, int W, int H, int C)
{
    Field f;
    f.p = (float *)p; f.W = W; f.H = H; f.C = C;
    f.pitch = round_up(W, 64);
    f.plane = (size_t)f.pitch * H;
    return f;
}

// LDS floats level l needs inside the bottom kernel: U and F planes, odd row pitch
static int bottom_pitch(const MGLevel &L) { return (L.g.x.n + 2) | 1; }
static long bottom_floats(const MGLevel &L) { return 2L * bottom_pitch(L) * (L.g.y.n + 2); }

// first level handled by the fused bottom kernel: the first l >= 1 from which all remaining
// levels fit the LDS budget together (never level 0: it runs the exact kernels)
static size_t bottom_start(Instance *I)
{
    for (size_t l = 1; l < I->mg.size(); ++l) {
        if (I->mg.size() - l > (size_t)MG_BOTTOM_MAX_LEVELS) continue;
        long tot = 0;
        for (size_t k = l; k < I->mg.size(); ++k) tot += bottom_floats(I->mg[k]);
        if (tot * (long)sizeof(float) <= (long)MG_BOTTOM_LDS_BYTES) return l;
    }
    return I->mg.size();
}

static int run_bottom(Instance *I, size_t l0, int pre, int post)
{
    MGBottomArgs a;
    a.nlevels = (int)(I->mg.size() - l0);
    a.pre = pre; a.post = post;
    const MGLevel &last = I->mg.back();
    a.coarse_sweeps = std::max(8, std::min(64, 2 * std::max(last.g.x.n, last.g.y.n)));
    int off = 0;
    for (int i = 0; i < a.nlevels; ++i) {
        const MGLevel &L = I->mg[l0 + i];
        a.lv[i].g = L.g; a.lv[i].omega = L.omega;
        a.lv[i].pitch = bottom_pitch(L);
        const int plane = a.lv[i].pitch * (L.g.y.n + 2);
        a.lv[i].offU = off; a.lv[i].offF = off + plane;
        off += 2 * plane;
    }
    a.lds_floats = off;
    a.Ftop = I->mg[l0].F;
    a.Utop = I->mg[l0].U;
    launch_mg_bottom(a, I->F.C, I->stream);
    return SC_OK;
}

static int build_levels(Instance *I)
{
    const int W = I->F.W, H = I->F.H, C = I->F.C;
    if (!I->mg.empty() && I->mg[0].F.p == I->F.p && I->mg[0].F.W == W && I->mg[0].F.H == H && I->mg[0].F.C == C)
        return SC_OK;
    I->mg.clear();
    struct L1 { int nx, ny; double ax, ay; };
    std::vector<L1> ls;
    ls.push_back({ W - 2, H - 2, 1.0, 1.0 });
    while (std::min(ls.back().nx, ls.back().ny) > 3) {
        L1 c;
        coarsen_1d(ls.back().nx, ls.back().ax, c.nx, c.ax);
        coarsen_1d(ls.back().ny, ls.back().ay, c.ny, c.ay);
        if (c.nx < 1 || c.ny < 1) break;
        ls.push_back(c);
    }
    const size_t nl = ls.size();
    if (I->mg_bufs.size() < 3 * nl) I->mg_bufs.resize(3 * nl);
    I->mg.resize(nl);
    for (size_t l = 0; l < nl; ++l) {
        MGLevel &L = I->mg[l];
        const int ncx = (l + 1 < nl) ? ls[l + 1].nx : 0, ncy = (l + 1 < nl) ? ls[l + 1].ny : 0;
        L.g.x = make_dim(ls[l].nx, ls[l].ax, ncx);
        L.g.y = make_dim(ls[l].ny, ls[l].ay, ncy);
        const double rho = 0.5 * (std::cos(M_PI / (ls[l].nx + 1.0)) + std::cos(M_PI / (ls[l].ny + 1.0)));
        L.omega = (float)(2.0 / (1.0 + std::sqrt(std::max(0.0, 1.0 - rho * rho))));
        if (l == 0) continue; // level 0 aliases the instance fields, bound per cycle
        const int Wl = ls[l].nx + 2, Hl = ls[l].ny + 2;
        Field proto = level_field(nullptr, Wl, Hl, C);
        for (int k = 0; k < 3; ++k) {
            int rc = ensure(I, I->mg_bufs[3 * l + k], proto.bytes() + 4096);
            if (rc) return rc;
        }
        L.U = level_field(I->mg_bufs[3 * l + 0].p, Wl, Hl, C);
        L.F = level_field(I->mg_bufs[3 * l + 1].p, Wl, Hl, C);
        L.T = level_field(I->mg_bufs[3 * l + 2].p, Wl, Hl, C);
        // rings and pads of F/U must be zero; ensure() zero-fills fresh memory, but a reused
        // larger buffer may hold stale data from another ROI size
        SC_HIP(I, hipMemsetAsync(L.U.p, 0, L.U.bytes(), I->stream));
        SC_HIP(I, hipMemsetAsync(L.F.p, 0, L.F.bytes(), I->stream));
        SC_HIP(I, hipMemsetAsync(L.T.p, 0, L.T.bytes(), I->stream));
    }
    I->mg[0].F = I->F;
    I->mg_bottom = bottom_start(I);
    return SC_OK;
}

// Smoothing of a coarse level (l >= 1) with the fused general kernel; U <-> T ping-pong, both
// carry zero rings.  mode (first launch only): TBM_ZEROIN = the current correction is all zero
// (nothing is read), TBM_PROLONG = add the interpolated correction of level l+1 while loading.
static int smooth_gen(Instance *I, size_t l, int n, int mode, Field E)
{
    MGLevel &L = I->mg[l];
    int left = n;
    while (left > 0) {
        const int T = std::min(2, left);
        if (!launch_rb_tb_gen(L.U, L.T, L.F, T, L.g, mode, E, I->stream)) break;
        std::swap(L.U, L.T);
        mode = TBM_PLAIN;
        left -= T;
    }
    if (left > 0 && mode != TBM_PLAIN) return SC_ERR_BAD_ARG; // cannot happen: depths 1 and 2 always exist
    for (int s = 0; s < left; ++s) {
        launch_rb_half_gen(L.U, L.F, 0, 1.0f, L.g, I->stream);
        launch_rb_half_gen(L.U, L.F, 1, 1.0f, L.g, I->stream);
    }
    return SC_OK;
}

static int vcycle(Instance *I, size_t l, int pre, int post)
{
    MGLevel &L = I->mg[l];
    int rc;
    if (l > 0 && l == I->mg_bottom) return run_bottom(I, l, pre, post);
    if (l + 1 == I->mg.size()) { // coarsest level outside the bottom kernel: SOR with its optimal factor
        const int n = std::max(8, std::min(64, 2 * std::max(L.g.x.n, L.g.y.n)));
        if (l == 0) return run_sweeps(I, SC_METHOD_SOR, n, L.omega, 1);
        launch_fill_zero(L.U, I->stream);
        for (int s = 0; s < n; ++s) {
            launch_rb_half_gen(L.U, L.F, 0, L.omega, L.g, I->stream);
            launch_rb_half_gen(L.U, L.F, 1, L.omega, L.g, I->stream);
        }
        return SC_OK;
    }
    MGLevel &Lc = I->mg[l + 1];
    // ---- pre-smoothing (levels >= 1 start from a zero correction), residual + restriction
    bool restricted = false;
    if (l == 0) {
        if ((rc = run_sweeps(I, SC_METHOD_RBGS, pre, 1.0f, I->opts.sweeps_per_launch))) return rc;
    } else if (pre > 0 && I->opts.sweeps_per_launch != 1 &&
               launch_cycle_coarse(L.T, L.F, Lc.F, L.g, pre, I->stream)) {
        std::swap(L.U, L.T);      // one launch did all three
        restricted = true;
    } else if (pre > 0) {
        if ((rc = smooth_gen(I, l, pre, TBM_ZEROIN, Field{}))) return rc;
    } else {
        launch_fill_zero(L.U, I->stream);
    }
    if (!restricted) launch_residual_restrict(l == 0 ? result(I) : L.U, L.F, Lc.F, L.g, I->stream);
    if ((rc = vcycle(I, l + 1, pre, post))) return rc;
    // ---- prolongation fused into the first post-smoothing launch
    if (l == 0) {
        const int T = std::min(2, post);
        int nb = 0;
        if (T > 0 && I->opts.sweeps_per_launch != 1) {
            Field &in = result(I);
            Field &out = I->result_in_U1 ? I->U0 : I->U1;
            nb = launch_rb_tb_prolong0(in, out, I->F, T, L.g, Lc.U, (float *)I->mg_partial.p, I->stream);
        }
        if (nb > 0) {
            I->result_in_U1 = !I->result_in_U1;
            I->info.sweep_launches += 1;
            launch_max_final((const float *)I->mg_partial.p, nb, I->d_maxcorr, I->stream);
            if ((rc = run_sweeps(I, SC_METHOD_RBGS, post - T, 1.0f, I->opts.sweeps_per_launch))) return rc;
        } else {
            launch_prolong_add(Lc.U, result(I), L.g, (float *)I->mg_partial.p, I->d_maxcorr, I->stream);
            if ((rc = run_sweeps(I, SC_METHOD_RBGS, post, 1.0f, I->opts.sweeps_per_launch))) return rc;
        }
    } else if (post > 0) {
        if ((rc = smooth_gen(I, l, post, TBM_PROLONG, Lc.U))) return rc;
    } else {
        launch_prolong_add(Lc.U, L.U, L.g, nullptr, nullptr, I->stream);
    }
    return SC_OK;
}

int mg_solve(Instance *I)
{
    const sc_solver_opts &o = I->opts;
    int rc = build_levels(I);
    if (rc) return rc;
    {
        const int nb = std::max(std::max(prolong_blocks(I->F.W - 2, I->F.H - 2, I->F.C), tb_blocks_level0(I->F.W, I->F.H, I->F.C, 1)),
                                cycle0_blocks(I->F.W, I->F.H, I->F.C, 1));
        if ((rc = ensure(I, I->mg_partial, sizeof(float) * (size_t)nb))) return rc;
    }
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    const float utol = o.update_tol > 0.f ? o.update_tol : 0.25f;
    const int budget = o.max_sweeps > 0 ? o.max_sweeps : 30;
    // the level-0 scratch is the ping-pong partner of the solution; the residual field only
    // writes its interior, and both buffers carry the same ring, so it stays a valid partner
    int cyc = 0;
    bool ok = false;
    // Fused level-0 form: one launch per cycle does [prolongation +] post-smoothing of this cycle,
    // pre-smoothing of the next, residual and restriction (sc_cycle0.hip).  The first launch has no
    // correction to add; after the last one the field has simply had `pre` extra sweeps.
    const bool fused0 = o.sweeps_per_launch != 1 && I->mg.size() >= 2 && pre >= 1 && pre <= 2 && post >= 1 &&
                        pre + post <= 4;
    if (fused0) {
        Field none{};
        launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, none, I->mg[0].g, pre, false, nullptr,
                      I->stream);
        I->result_in_U1 = !I->result_in_U1;
        I->info.sweep_launches += 1;
        while (cyc < budget) {
            if ((rc = vcycle(I, 1, pre, post))) return rc;
            const int nb = launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U,
                                         I->mg[0].g, post + pre, true, (float *)I->mg_partial.p, I->stream);
            if (nb <= 0) { I->err = "cycle0: unsupported depth"; return SC_ERR_BAD_ARG; }
            I->result_in_U1 = !I->result_in_U1;
            I->info.sweep_launches += 1;
            launch_max_final((const float *)I->mg_partial.p, nb, I->d_maxcorr, I->stream);
            ++cyc;
            SC_HIP(I, hipGetLastError());
            // The first two corrections of a solve are never below the stop threshold unless the
            // initial guess was already the answer, and every check costs a host round trip
            // (~35 us of idle GPU), so checking starts with the third cycle.
            if (cyc < 3 && cyc < budget && o.tol <= 0.f) continue;
            SC_HIP(I, hipMemcpyAsync(I->h_maxcorr, I->d_maxcorr, sizeof(unsigned), hipMemcpyDeviceToHost, I->stream));
            SC_HIP(I, hipStreamSynchronize(I->stream));
            float m;
            unsigned bits = *I->h_maxcorr;
            memcpy(&m, &bits, sizeof(float));
            I->info.last_update = m;
            if (o.tol > 0.f) {
                double r[2];
                if ((rc = eval_residual(I, r))) return rc;
                const double rel = (r[1] > 0.0) ? std::sqrt(r[0] / r[1]) : std::sqrt(r[0]);
                I->info.rel_residual = rel;
                if (rel <= (double)o.tol) { ok = true; break; }
            }
            if (m <= utol) { ok = true; break; }
        }
        I->info.sweeps = cyc;
        I->info.converged = ok ? 1 : 0;
        return ok ? SC_OK : SC_ERR_NOT_CONVERGED;
    }
    while (cyc < budget) {
        if ((rc = vcycle(I, 0, pre, post))) return rc;
        ++cyc;
        SC_HIP(I, hipGetLastError());
        if (I->mg.size() == 1) { ok = true; break; } // single level: solved by SOR above
        SC_HIP(I, hipMemcpyAsync(I->h_maxcorr, I->d_maxcorr, sizeof(unsigned), hipMemcpyDeviceToHost, I->stream));
        SC_HIP(I, hipStreamSynchronize(I->stream));
        float m;
        unsigned bits = *I->h_maxcorr;
        memcpy(&m, &bits, sizeof(float));
        I->info.last_update = m;
        if (o.tol > 0.f) { // optional residual-based stop
            double r[2];
            if ((rc = eval_residual(I, r))) return rc;
            const double rel = (r[1] > 0.0) ? std::sqrt(r[0] / r[1]) : std::sqrt(r[0]);
            I->info.rel_residual = rel;
            if (rel <= (double)o.tol) { ok = true; break; }
        }
        if (m <= utol) { ok = true; break; }
    }
    I->info.sweeps = cyc;
    I->info.converged = ok ? 1 : 0;
    return ok ? SC_OK : SC_ERR_NOT_CONVERGED;
}

} // namespace sc
