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

// first level handled by the fused bottom kernel (never level 0: it uses the exact kernels)
static size_t bottom_start(Instance *I)
{
    for (size_t l = 1; l < I->mg.size(); ++l)
        if ((long)I->mg[l].g.x.n * I->mg[l].g.y.n <= MG_BOTTOM_POINTS && I->mg.size() - l <= MG_BOTTOM_MAX_LEVELS)
            return l;
    return I->mg.size();
}

static int run_bottom(Instance *I, size_t l0, int pre, int post)
{
    MGBottomArgs a;
    a.nlevels = (int)(I->mg.size() - l0);
    a.pre = pre; a.post = post;
    const MGLevel &last = I->mg.back();
    a.coarse_sweeps = std::max(8, std::min(64, 2 * std::max(last.g.x.n, last.g.y.n)));
    for (int i = 0; i < a.nlevels; ++i) {
        const MGLevel &L = I->mg[l0 + i];
        a.lv[i].U = L.U; a.lv[i].F = L.F; a.lv[i].T = L.T; a.lv[i].g = L.g; a.lv[i].omega = L.omega;
    }
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

static int smooth(Instance *I, size_t l, int n)
{
    if (n <= 0) return SC_OK;
    if (l == 0) return run_sweeps(I, SC_METHOD_RBGS, n, 1.0f, I->opts.sweeps_per_launch);
    MGLevel &L = I->mg[l];
    for (int s = 0; s < n; ++s) {
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
    if (l + 1 == I->mg.size()) { // coarsest: SOR with the level's optimal factor
        const int n = std::max(8, std::min(64, 2 * std::max(L.g.x.n, L.g.y.n)));
        if (l == 0) return run_sweeps(I, SC_METHOD_SOR, n, L.omega, 1);
        for (int s = 0; s < n; ++s) {
            launch_rb_half_gen(L.U, L.F, 0, L.omega, L.g, I->stream);
            launch_rb_half_gen(L.U, L.F, 1, L.omega, L.g, I->stream);
        }
        return SC_OK;
    }
    if ((rc = smooth(I, l, pre))) return rc;
    MGLevel &Lc = I->mg[l + 1];
    Field Ul = (l == 0) ? result(I) : L.U;
    Field Tl = (l == 0) ? (I->result_in_U1 ? I->U0 : I->U1) : L.T;
    launch_residual_field(Ul, L.F, Tl, L.g, I->stream);
    launch_restrict(Tl, Lc.F, L.g, I->stream);
    if (l + 1 != I->mg_bottom) launch_fill_zero(Lc.U, I->stream); // the bottom kernel zeroes its own top level
    if ((rc = vcycle(I, l + 1, pre, post))) return rc;
    Ul = (l == 0) ? result(I) : L.U;
    launch_prolong_add(Lc.U, Ul, L.g, l == 0 ? (float *)I->mg_partial.p : nullptr, l == 0 ? I->d_maxcorr : nullptr,
                       I->stream);
    if ((rc = smooth(I, l, post))) return rc;
    return SC_OK;
}

int mg_solve(Instance *I)
{
    const sc_solver_opts &o = I->opts;
    int rc = build_levels(I);
    if (rc) return rc;
    if ((rc = ensure(I, I->mg_partial, sizeof(float) * (size_t)prolong_blocks(I->F.W - 2, I->F.H - 2, I->F.C)))) return rc;
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    const float utol = o.update_tol > 0.f ? o.update_tol : 0.02f;
    const int budget = o.max_sweeps > 0 ? o.max_sweeps : 30;
    // the level-0 scratch is the ping-pong partner of the solution; the residual field only
    // writes its interior, and both buffers carry the same ring, so it stays a valid partner
    int cyc = 0;
    bool ok = false;
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
