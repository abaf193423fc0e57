// sc_multigrid.cpp -- geometric multigrid V-cycle for the ROI Poisson system (SURVEY.md
// section 8 row f1): red-black GS smoothing, residual in double, normalised-transpose
// restriction, bilinear prolongation.  Arbitrary ROI sizes coarsen by letting the LAST grid
// interval of each level differ from the others (MGDim), so the Dirichlet ring never moves.
// Level 0 runs the exact 5-point kernels of the sweep solvers; coarser levels the general
// ones.  Converges ~20x per V(2,2) cycle at every size tried (tools/mg_proto2.py).
#include "sc_instance.h"
#include "sc_fd_closed.h"
#include <algorithm>
#include <cmath>
#include <functional>
#include <cstring>
#include <cstdlib>

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
    // at least two pad columns behind the ring: the level-0 kernel reads three coarse columns starting at an even column <= nc
    // (k_cycle0's prolongation), and must find them where it expects them, not shifted by an address clamp
    f.pitch = round_up(W + 2, 64);
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

// ---------------------------------------------------------------------------------------------
// Direct solve of the bottom's first level(s) by fast diagonalisation.
// A level's operator is  (A u)[y][x] = sum_x' Tx[x][x'] u[y][x'] + sum_y' Ty[y][y'] u[y'][x]  with
// tridiagonal 1-D parts: rows (1, -2, 1), last row (cw_last, -d_last) (MGDim).  T is not symmetric
// (the last sub-diagonal is cw_last, the super-diagonal above it 1) but E T E^-1 is, with
// E = diag(1, .., 1, 1/sqrt(cw_last)); its eigen-decomposition Q L Q^T gives T = V L V^-1 with
// V = E^-1 Q, V^-1 = Q^T E.  Everything here is double; the device gets float matrices.
// ---------------------------------------------------------------------------------------------
// Eigen-decomposition of a symmetric tridiagonal matrix by implicit QL with Wilkinson shifts.
// d: diagonal (n) -> eigenvalues; e: sub-diagonal, e[i] couples i and i+1 (n-1 used, e[n-1] = 0);
// zt: n x n, row k = eigenvector k on return (kept transposed so the rotation loop is contiguous).
static bool tridiag_ql(int n, std::vector<double> &d, std::vector<double> &e, std::vector<double> &zt)
{
    zt.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) zt[(size_t)i * n + i] = 1.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 1.1e-16 * dd) break;
            }
            if (m != l) {
                if (++iter > 80) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    double *zi = &zt[(size_t)i * n], *zi1 = &zt[(size_t)(i + 1) * n];
                    for (int k = 0; k < n; ++k) {
                        f = zi1[k];
                        zi1[k] = s * zi[k] + c * f;
                        zi[k] = c * zi[k] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

static bool fd_decompose(const MGDim &g, FD1 &o)
{
    const int n = g.n;
    o.n = n; o.cw_last = g.cw_last; o.d_last = g.d_last;
    o.ee.assign(n, 1.0);
    std::vector<double> d(n, -2.0), e(n, 0.0);
    for (int i = 0; i + 1 < n; ++i) e[i] = 1.0;
    d[n - 1] = -(double)g.d_last;
    if (n >= 2) {
        e[n - 2] = std::sqrt((double)g.cw_last);          // sqrt(sub * super) = sqrt(cw_last * 1)
        o.ee[n - 1] = 1.0 / std::sqrt((double)g.cw_last);
    }
    if (!tridiag_ql(n, d, e, o.q)) return false;
    o.lam = d;
    return true;
}

// host-only check of the decomposition (sc_hip_selftest_host): max |T v_k - l_k v_k| and max |V^-1 V - I| over a few
// level operators, regular and with an irregular last interval
double fd_selftest_error()
{
    double worst = 0.0;
    const int ns[] = { 1, 2, 3, 7, 31, 63, 74, 128 };
    const double alphas[] = { 1.0, 0.5, 0.75, 1.5, 0.96875 };
    for (int n : ns)
        for (double a : alphas) {
            MGDim g = make_dim(n, a, 0);
            FD1 f;
            if (!fd_decompose(g, f)) return 1e30;
            auto T = [&](int i, int j) -> double {          // the level operator itself
                if (i == j) return i == n - 1 ? -(double)g.d_last : -2.0;
                if (j == i + 1) return 1.0;
                if (j == i - 1) return i == n - 1 ? (double)g.cw_last : 1.0;
                return 0.0;
            };
            for (int k = 0; k < n; ++k) {
                for (int i = 0; i < n; ++i) {
                    double tv = 0.0;
                    for (int j = std::max(0, i - 1); j <= std::min(n - 1, i + 1); ++j) tv += T(i, j) * f.q[(size_t)k * n + j] / f.ee[j];
                    worst = std::max(worst, std::fabs(tv - f.lam[k] * f.q[(size_t)k * n + i] / f.ee[i]));
                }
                for (int m = 0; m < n; ++m) {                // rows of V^-1 = Q^T E against columns of V = E^-1 Q
                    double dot = 0.0;
                    for (int i = 0; i < n; ++i) dot += f.q[(size_t)k * n + i] * f.ee[i] * f.q[(size_t)m * n + i] / f.ee[i];
                    worst = std::max(worst, std::fabs(dot - (k == m ? 1.0 : 0.0)));
                }
            }
        }
    return worst;
}

// Chooses the bottom level solved directly and has its matrices built ON THE DEVICE from the closed-form eigenpairs of the
// level's two 1-D operators (sc_fd_closed.h, k_fd_build) -- no host eigen-solve, no staging copy, no wait.  The build runs on
// the instance's second stream, beside the first launches of the clone that needs it; run_bottom() makes the main stream
// wait for it.  (Rounds 1-3 ran the implicit-QL solve above on the host for every new ROI size: 0.1-0.3 ms per direction at
// n = 63, as long as the clone itself; it now only serves sc_hip_selftest_host as the reference the closed form is checked
// against.)  I->fd_level = -1 when nothing fits.
static int build_fd(Instance *I)
{
    I->fd_level = -1;
    if ((I->opts.flags & SC_FLAG_VCYCLE_BOTTOM) || I->mg_bottom >= I->mg.size()) return SC_OK;
    long planes = 0;
    I->fd_mm = false;
    for (size_t l = I->mg_bottom; l < I->mg.size(); ++l) {
        const MGLevel &L = I->mg[l];
        planes += bottom_floats(L);
        const int nx = L.g.x.n, ny = L.g.y.n, nxp = round_up(nx, 4), nyp = round_up(ny, 4);
        const int dmax = I->opts.mg_direct_max > 0 ? std::min(I->opts.mg_direct_max, 128) : SC_MG_DIRECT_MAX_DEFAULT;
        if (nx > dmax || ny > dmax) continue;
        // the bottom's first level on the matrix cores (k_mg_bottom_mm): up to 96 unknowns per side, no LDS budget to meet
        const bool mm = l == I->mg_bottom && nx <= 96 && ny <= 96 && !legacy_path(I->opts, SC_LEGACY_BOTTOM_F32);
        if (!mm && (planes + fd_lds_floats(nxp, nyp)) * (long)sizeof(float) > (long)MG_BOTTOM_LDS_BYTES) continue;
        const long nf = (fd_mat_floats(nxp, nyp) + 15) & ~15L;          // the matrix-core operands behind the float matrices, 64-byte aligned
        const int NPX = round_up(nx, 32), NPY = round_up(ny, 32);
        int rc;
        if (I->fd_pending) {          // a build nobody waited for (a solve that never reached its bottom): order it in front of whatever
            SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_fd, 0));      // follows on the main stream -- ensure() below waits for that stream before it frees
            I->fd_pending = false;
        }
        if ((rc = ensure(I, I->mg_fd, sizeof(float) * (size_t)nf + (mm ? (size_t)fd_mm_bytes(NPX, NPY) : 0)))) return rc;
        // everything that read the previous matrices has been enqueued on the main stream: the build starts behind it
        SC_HIP(I, hipEventRecord(I->ev_fd_fork, I->stream));
        SC_HIP(I, hipStreamWaitEvent(I->aux, I->ev_fd_fork, 0));
        launch_fd_build((float *)I->mg_fd.p, L.g, nxp, nyp, I->aux, mm ? (unsigned char *)((float *)I->mg_fd.p + nf) : nullptr, NPX, NPY);
        SC_HIP(I, hipGetLastError());
        SC_HIP(I, hipEventRecord(I->ev_fd, I->aux));
        I->fd_pending = true;
        I->fd_level = (int)(l - I->mg_bottom);
        I->fd_nxp = nxp; I->fd_nyp = nyp;
        I->fd_mm = mm; I->fd_npx = NPX; I->fd_npy = NPY; I->fd_mm_off = (size_t)nf;
        return SC_OK;
    }
    return SC_OK;
}

// host-only check of the closed form (sc_hip_selftest_host): its matrices V, V^-1 and eigenvalues against the QL-based
// decomposition over level operators of every size the bottom solve can meet, regular and with an irregular last interval
// on either side of the alpha = 0.7071 threshold (one eigenvalue below -4).  Returns the worst deviation found.
double fd_closed_selftest_error()
{
    double worst = 0.0;
    const double alphas[] = { 1.0, 0.5, 0.625, 0.70703125, 0.7109375, 0.75, 0.875, 1.125, 1.25, 1.5, 0.96875 };
    for (int n = 1; n <= 128; n += (n < 20 ? 1 : 9))
        for (double a : alphas) {
            MGDim g = make_dim(n, a, 0);
            FD1 f;
            if (!fd_decompose(g, f)) return 1e30;
            std::vector<FdPair> p(n);
            for (int k = 0; k < n; ++k) p[k] = fd_pair(k, n, (double)g.cw_last, (double)g.d_last);
            // eigenvalues: the two sets must agree as sets (QL's order is arbitrary)
            std::vector<double> la(f.lam), lb(n);
            for (int k = 0; k < n; ++k) lb[k] = p[k].lam;
            std::sort(la.begin(), la.end()); std::sort(lb.begin(), lb.end());
            for (int k = 0; k < n; ++k) worst = std::max(worst, std::fabs(la[k] - lb[k]));
            // T v = lambda v for the closed form's own vectors, and V^-1 V = I
            auto T = [&](int i, int j) -> double {
                if (i == j) return i == n - 1 ? -(double)g.d_last : -2.0;
                if (j == i + 1) return 1.0;
                if (j == i - 1) return i == n - 1 ? (double)g.cw_last : 1.0;
                return 0.0;
            };
            std::vector<double> V((size_t)n * n), Vi((size_t)n * n);      // V[x][k], Vinv[k][x]
            for (int k = 0; k < n; ++k)
                for (int x = 0; x < n; ++x) {
                    const double v = fd_component(p[k], x + 1, n) * p[k].inv_norm;
                    V[(size_t)x * n + k] = v;
                    Vi[(size_t)k * n + x] = v * (x == n - 1 ? 1.0 / (double)g.cw_last : 1.0);
                }
            for (int k = 0; k < n; ++k) {
                for (int i = 0; i < n; ++i) {
                    double tv = 0.0;
                    for (int j = std::max(0, i - 1); j <= std::min(n - 1, i + 1); ++j) tv += T(i, j) * V[(size_t)j * n + k];
                    worst = std::max(worst, std::fabs(tv - p[k].lam * V[(size_t)i * n + k]));
                }
                for (int m = 0; m < n; ++m) {
                    double dot = 0.0;
                    for (int x = 0; x < n; ++x) dot += Vi[(size_t)k * n + x] * V[(size_t)x * n + m];
                    worst = std::max(worst, std::fabs(dot - (k == m ? 1.0 : 0.0)));
                }
            }
        }
    return worst;
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
    a.fd_level = I->fd_level; a.fd_nxp = I->fd_nxp; a.fd_nyp = I->fd_nyp;
    a.fd_mats = (const float *)I->mg_fd.p;
    a.fd_off = 0;
    if (a.fd_level >= 0) {
        // levels below the directly solved one are not visited: the FD region takes their place in LDS
        a.fd_off = (a.lv[a.fd_level].offF + a.lv[a.fd_level].pitch * (I->mg[l0 + a.fd_level].g.y.n + 2) + 3) & ~3;
        off = a.fd_off + (int)fd_lds_floats(a.fd_nxp, a.fd_nyp);
    }
    a.lds_floats = off;
    a.Ftop = I->mg[l0].F;
    a.Utop = I->mg[l0].U;
    if (I->fd_pending) {          // the matrices of a new hierarchy are being built on the second stream (build_fd)
        SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_fd, 0));
        I->fd_pending = false;
    }
    if (I->fd_mm && I->fd_level == 0) {      // the usual case: this level solved directly on the matrix cores, nothing below it is visited
        MGBottomMM m;
        m.mm = (const unsigned char *)((const float *)I->mg_fd.p + I->fd_mm_off);
        m.Ftop = I->mg[l0].F; m.Utop = I->mg[l0].U;
        m.nx = I->mg[l0].g.x.n; m.ny = I->mg[l0].g.y.n;
        if (launch_mg_bottom_mm(m, I->fd_npx, I->fd_npy, I->F.C, I->stream)) return SC_OK;
    }
    launch_mg_bottom(a, I->F.C, I->stream);
    return SC_OK;
}

// The level above the bottom and the bottom in one launch (k_mg_tail): level l is that level, the bottom's first level is the one
// solved directly on the matrix cores with at most 64 padded unknowns per side, and l itself is a plain float level (>= 2: level 1
// has its own composed / float16 forms).
static bool tail_serves(const Instance *I, size_t l)
{
    if (l < 2 || l + 1 != I->mg_bottom || legacy_path(I->opts, SC_LEGACY_SEPARATE_TAIL) || I->opts.sweeps_per_launch == 1) return false;
    if (!I->fd_mm || I->fd_level != 0 || I->fd_npx > 64 || I->fd_npy > 64) return false;
    if ((I->opts.mg_pre > 0 ? I->opts.mg_pre : 2) < 1) return false;          // the launch takes the residual of the colour swept last as zero
    const MGGeom &g = I->mg[l].g;
    return g.x.n <= 127 && g.y.n <= 127 && g.x.nc <= 63 && g.y.nc <= 63;
}

static int run_tail(Instance *I, size_t l, int pre, int post, bool &done, unsigned long long *stamps = nullptr)
{
    done = false;
    MGTail t;
    t.stamps = stamps;
    t.mm = (const unsigned char *)((const float *)I->mg_fd.p + I->fd_mm_off);
    t.F = I->mg[l].F; t.U = I->mg[l].U; t.g = I->mg[l].g; t.pre = pre; t.post = post;
    t.rag = I->rag.dev; t.lev = (int)l; t.rag_uniform = I->rag.pad_uniform;
    if (I->fd_pending) {
        SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_fd, 0));
        I->fd_pending = false;
    }
    done = launch_mg_tail(t, I->fd_npx, I->fd_npy, I->F.C, I->stream);
    return SC_OK;
}

// The ladder of levels of a W x H field (ring included): the geometry of every level (and of its transfer to the next coarser one).
// Host arithmetic only; the size-class planner (sc_ragged.cpp) runs it per member.
void mg_plan_levels(int W, int H, std::vector<MGGeom> &g)
{
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
    g.resize(nl);
    for (size_t l = 0; l < nl; ++l) {
        const int ncx = (l + 1 < nl) ? ls[l + 1].nx : 0, ncy = (l + 1 < nl) ? ls[l + 1].ny : 0;
        g[l].x = make_dim(ls[l].nx, ls[l].ax, ncx);
        g[l].y = make_dim(ls[l].ny, ls[l].ay, ncy);
    }
}

// The default hierarchy's deepest launched level (see build_levels): the first level >= 2 with at most 127 unknowns per side, held in
// registers by k_mg_tail with the level below it solved directly in the same launch; 0: this ladder ends differently (its level 1
// is solved directly -- at most 64 unknowns per side: 10-13 us per solve for a group of sixteen, against ~24 for a level-1 launch plus
// k_mg_tail; up to 96 until late in round 5, but the 96-wide solve takes 31-34 us (ROIs of 131..194 pixels: measured on groups of
// 16, tools/class_timeline.sh) --, or no such level exists)
size_t mg_default_tail_level(const std::vector<MGGeom> &g)
{
    const size_t nl = g.size();
    size_t a = 0;
    for (size_t l = 2; l + 1 < nl && !a; ++l)
        if (g[l].x.n <= 127 && g[l].y.n <= 127) a = l;
    const bool level1_direct = nl > 1 && g[1].x.n <= 64 && g[1].y.n <= 64;
    return (a && !(a == 2 && level1_direct)) ? a : 0;
}

static int build_levels(Instance *I)
{
    if (I->rag.dev) {          // a size class: rag_begin_builds built the hierarchy (mg_build_levels_rag)
        if (!I->rag.levels_built || I->mg.empty()) { I->err = "size class: hierarchy missing"; return SC_ERR_BAD_ARG; }
        return SC_OK;
    }
    const int W = I->F.W, H = I->F.H, C = I->F.C;
    if (!I->mg.empty() && I->mg[0].F.p == I->F.p && I->mg[0].F.W == W && I->mg[0].F.H == H && I->mg[0].F.C == C)
        return SC_OK;
    I->info.new_size = 1;
    I->mg.clear();
    std::vector<MGGeom> plan;
    mg_plan_levels(W, H, plan);
    const size_t nl = plan.size();
    if (I->mg_bufs.size() < 3 * nl) I->mg_bufs.resize(3 * nl);
    I->mg.resize(nl);
    ZeroJobs zj{};
    for (size_t l = 0; l < nl; ++l) {
        MGLevel &L = I->mg[l];
        L.g = plan[l];
        const double rho = 0.5 * (std::cos(M_PI / (L.g.x.n + 1.0)) + std::cos(M_PI / (L.g.y.n + 1.0)));
        L.omega = (float)(2.0 / (1.0 + std::sqrt(std::max(0.0, 1.0 - rho * rho))));
        if (l == 0) continue; // level 0 aliases the instance fields, bound per cycle
        const int Wl = L.g.x.n + 2, Hl = L.g.y.n + 2;
        Field proto = level_field(nullptr, Wl, Hl, C);
        for (int k = 0; k < 3; ++k) {
            int rc = ensure(I, I->mg_bufs[3 * l + k], proto.bytes() + 4096, false);      // (zeroed below, by the launch that zeroes every plane)
            if (rc) return rc;
        }
        L.U = level_field(I->mg_bufs[3 * l + 0].p, Wl, Hl, C);
        L.F = level_field(I->mg_bufs[3 * l + 1].p, Wl, Hl, C);
        L.T = level_field(I->mg_bufs[3 * l + 2].p, Wl, Hl, C);
        // rings and pads of F/U must be zero; ensure() zero-fills fresh memory, but a reused
        // larger buffer may hold stale data from another ROI size: every plane of every level in ONE launch below
        // (24-36 memsets were 70-100 us of launches in front of the first clone at a new size)
        for (const Field *f : { &L.U, &L.F, &L.T }) {
            if (zj.count == ZeroJobs::MAX) { launch_zero_multi(zj, I->stream); zj.count = 0; }
            zj.p[zj.count] = f->p; zj.n16[zj.count] = (f->bytes() + 15) / 16; ++zj.count;       // buffers are 4096 bytes larger than the field
        }
    }
    launch_zero_multi(zj, I->stream);
    SC_HIP(I, hipGetLastError());
    I->mg[0].F = I->F;
    I->mg_bottom = bottom_start(I);
    // Default hierarchy since round 4: the deepest launched level ("A") is the first one (>= 2) with at most 127 unknowns per side --
    // k_mg_tail holds it in registers -- and the level below it ("B", at most 63 per side) is the one solved directly, on the matrix
    // cores, inside the same launch.  The LDS-fit rule above chose the bottom in rounds 1-3; where it landed on a level with 97 .. ~190
    // unknowns on a side (ROIs like 2090 x 1632, 2500 x 1300, 3540^2: no matrix-core solve, an LDS-resident V-cycle inside
    // k_mg_bottom instead) a cycle cost 60 us more than at the sizes next to it (0.55 against 0.38 ms for one clone).  Kept: a ROI
    // whose level 1 fits the matrix-core solve at 64 (solved there; at 65..96 only where the ladder has no level for k_mg_tail),
    // the flags that ask for the older bottoms.
    if (!(I->opts.flags & SC_FLAG_VCYCLE_BOTTOM) && !legacy_path(I->opts, SC_LEGACY_BOTTOM_F32) && I->opts.mg_direct_max <= 0) {
        const size_t a = mg_default_tail_level(plan);
        const bool level1_direct = nl > 1 && I->mg[1].g.x.n <= 96 && I->mg[1].g.y.n <= 96;
        if (a) I->mg_bottom = a + 1;
        else if (!level1_direct)
            for (size_t l = 1; l < nl; ++l)
                if (I->mg[l].g.x.n <= 96 && I->mg[l].g.y.n <= 96) { I->mg_bottom = l; break; }
    }
    I->mg_l1_half = false;        // fresh planes: all zero in either format
    return build_fd(I);
}

// The hierarchy of a SIZE CLASS (RagState, sc_instance.h): level planes at the class's strides -- the largest width and height any
// member has on that level --, every plane zeroed (a member's ring and what lies beyond it must be zero, and the slot may have held a
// larger member a call ago), the members' bottom matrices by one launch on the second stream.  The per-member geometries are in the
// table on the device; I->mg[l].g holds the class's MAXIMA (grid sizes and the launchers' shape tests read those).
// Called from rag_begin_builds: the zeroing goes to `zero_on` (the instance's second stream, which the main stream joins in front of its
// first coarse-level launch, mg_solve) -- the caller has ordered that stream behind everything that read the planes before.
int mg_build_levels_rag(Instance *I, hipStream_t zero_on)
{
    RagState &R = I->rag;
    R.levels_built = false;
    const int C = I->F.C, n = R.n;
    const size_t nl = (size_t)R.nl;
    I->info.new_size = 1;
    I->mg.clear();
    if (I->mg_bufs.size() < 3 * nl) I->mg_bufs.resize(3 * nl);
    I->mg.resize(nl);
    ZeroJobs zj{};
    for (size_t l = 0; l < nl; ++l) {
        MGLevel &L = I->mg[l];
        L.g = R.host[0].g[l];
        for (int i = 1; i < n; ++i) {
            const MGGeom &g = R.host[i].g[l];
            L.g.x.n = std::max(L.g.x.n, g.x.n); L.g.x.nc = std::max(L.g.x.nc, g.x.nc);
            L.g.y.n = std::max(L.g.y.n, g.y.n); L.g.y.nc = std::max(L.g.y.nc, g.y.nc);
        }
        L.omega = 1.f;
        if (l == 0) continue;
        const int Wl = L.g.x.n + 2, Hl = L.g.y.n + 2;
        Field proto = level_field(nullptr, Wl, Hl, C);
        for (int k = 0; k < 3; ++k) {
            int rc = ensure(I, I->mg_bufs[3 * l + k], proto.bytes() + 4096, false);      // (zeroed below, by the launch that zeroes every plane)
            if (rc) return rc;
        }
        L.U = level_field(I->mg_bufs[3 * l + 0].p, Wl, Hl, C);
        L.F = level_field(I->mg_bufs[3 * l + 1].p, Wl, Hl, C);
        L.T = level_field(I->mg_bufs[3 * l + 2].p, Wl, Hl, C);
        // What must be zero: a member's ring and everything beyond it in the planes a finer level interpolates FROM -- U and its
        // ping-pong partner T (the launches write a member's own extent only, and the slot may have held a larger member a call
        // ago).  Right-hand sides are read under the interior masks only: F needs nothing.
        for (const Field *f : { &L.U, &L.T }) {
            if (zj.count == ZeroJobs::MAX) { launch_zero_multi(zj, zero_on); zj.count = 0; }
            zj.p[zj.count] = f->p; zj.n16[zj.count] = (f->bytes() + 15) / 16; ++zj.count;
        }
    }
    launch_zero_multi(zj, zero_on);
    SC_HIP(I, hipGetLastError());
    I->mg[0].F = I->F;
    I->mg_bottom = (size_t)R.tail + 1;
    I->mg_l1_half = true;      // a class runs the fast path (plan_size): float16 level 1, and its planes are all zero -- valid in either format, nothing to re-zero in mg_solve
    // the class's bottom: every member's level below `tail` solved directly on the matrix cores inside k_mg_tail, operands padded
    // alike; the matrices are being built on the third stream since rag_begin_builds (run_tail waits for them)
    I->fd_level = 0; I->fd_mm = true; I->fd_npx = R.npx; I->fd_npy = R.npy; I->fd_nxp = I->fd_nyp = 0; I->fd_mm_off = 0;
    R.levels_built = true;
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
        if (!launch_rb_tb_gen(L.U, L.T, L.F, T, L.g, mode, E, I->stream, I->rag.dev, (int)l)) {
            if (I->rag.dev) { I->err = "size class: coarse-level form not instantiated"; return SC_ERR_BAD_ARG; }
            break;
        }
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

// no_post: bit l set = level l gets no post-smoothing and no prolongation launch of its own -- the level above interpolates
// from "its correction + the interpolated correction of the level below" directly (sc_mg_device.h, ComposeArgs).  Used for
// level 1 (composed by the level-0 launch); doing the same for level 3 inside level 2's post launch was measured neutral
// (481 vs 494 us for a single 2048^2 clone, no change in throughput) and is not kept.
static int vcycle(Instance *I, size_t l, int pre, int post, unsigned no_post = 0)
{
    const bool skip_post = l > 0 && l < 32 && ((no_post >> l) & 1u);
    MGLevel &L = I->mg[l];
    int rc;
    if (l > 0 && l == I->mg_bottom) return run_bottom(I, l, pre, post);
    if (!skip_post && tail_serves(I, l)) {
        bool done = false;
        if ((rc = run_tail(I, l, pre, post, done)) || done) return rc;
        if (I->rag.dev) { I->err = "size class: the bottom launch does not serve this shape"; return SC_ERR_BAD_ARG; }
    }
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
    // a level without post-smoothing does all its sweeps before the restriction
    int pre_here = pre;
    if (skip_post) {
        const int want = I->opts.mg_level1_sweeps > 0 ? I->opts.mg_level1_sweeps : 4;   // measured (tests/tools/level1_sweeps.py): 3 sweeps are 5 % faster per cycle but full-range noise then needs a 4th cycle
        pre_here = std::max(pre, std::min(want, pre + post));
    }
    // ---- pre-smoothing (levels >= 1 start from a zero correction), residual + restriction
    bool restricted = false;
    if (l == 0) {
        if ((rc = run_sweeps(I, SC_METHOD_RBGS, pre_here, 1.0f, I->opts.sweeps_per_launch))) return rc;
    } else if (pre_here > 0 && I->opts.sweeps_per_launch != 1 &&
               launch_cycle_coarse(L.T, L.F, Lc.F, L.g, pre_here, I->stream, l == 1 && skip_post && mg_level1_half(I), I->rag.dev, (int)l)) {
        std::swap(L.U, L.T);      // one launch did all three
        restricted = true;
    } else if (I->rag.dev) {
        I->err = "size class: coarse-level form not instantiated"; return SC_ERR_BAD_ARG;
    } else if (pre_here > 0) {
        if ((rc = smooth_gen(I, l, pre_here, TBM_ZEROIN, Field{}))) return rc;
    } else {
        launch_fill_zero(L.U, I->stream);
    }
    if (!restricted) launch_residual_restrict(l == 0 ? result(I) : L.U, L.F, Lc.F, L.g, I->stream);
    if ((rc = vcycle(I, l + 1, pre, post, no_post))) return rc;
    if (skip_post) return SC_OK;
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

// The fused level-0 cycle kernel is the only reader of the right-hand side that understands float16; every other
// path (sweep solvers, unfused cycle, residual-based stop rule, stage hooks) needs the float field.
static bool fused_level0(const sc_solver_opts &o)
{
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    return o.sweeps_per_launch != 1 && pre >= 1 && pre <= 2 && post >= 1 && post <= 2;   // the forms sc_cycle0.hip instantiates
}

// Fused solve on the current hierarchy: does level 1 run pre-smoothing only, with the level-0 launch composing its
// prolongation source from levels 1 and 2 (sc_cycle0.hip, ComposeArgs)?  Needs a launched level 1 with a level 2 below
// it and the standard 2 + 2 cycle.  The contraction per cycle is within a few percent of the full V(2,2)
// (oracle/mg_np.py runs the same schedule); one launch per cycle less is worth ~10 % of the clone throughput.
bool mg_composes_level1(const Instance *I)
{
    const sc_solver_opts &o = I->opts;
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    return !(o.flags & SC_FLAG_NO_COMPOSE_L1) && I->mg.size() >= 3 && I->mg_bottom >= 2 && pre == 2 && post == 2;
}

// Level 1 with float16 fields (sc_cycle0.hip, TAG bit 7): the composed schedule on the float16 right-hand side -- the default
// fast path -- with level 1's standard four sweeps (the only depth the float16 level-1 launch is instantiated for).
bool mg_level1_half(const Instance *I)
{
    const sc_solver_opts &o = I->opts;
    return !(o.flags & SC_FLAG_FLOAT_L1) && mg_composes_level1(I) && I->f_half && fused_level0(o) && o.tol <= 0.f &&
           (o.mg_level1_sweeps == 0 || o.mg_level1_sweeps == 4);
}

// The field between the FIRST level-0 launches of a solve as 16-bit fixed point (sc_cycle0.hip, TAG bits 8, 9): the fast path
// with float16 level-1 fields whose last cycle leaves output bytes (`out_wanted` in mg_solve).  The launch before the judged
// cycle reads the 16-bit field and writes float: the judged cycle, and a solve that goes on after it, run on float fields, and
// two cycles lie between the last rounding (<= 1/128) and the output.
static bool mg_field_q16(const Instance *I, bool out_wanted)
{
    return out_wanted && I->u_half && !(I->opts.flags & SC_FLAG_FLOAT_FIELD) && !I->force_float_field && mg_level1_half(I);
}

bool mg_reads_half_rhs(const Instance *I)
{
    const sc_solver_opts &o = I->opts;
    // at least two levels: min(W, H) - 2 > 3 (build_levels)
    return !(o.flags & (SC_FLAG_FLOAT_RHS | SC_FLAG_OPENCV_GREY_MASK)) && effective_method(I) == SC_METHOD_MULTIGRID && o.tol <= 0.f && fused_level0(o) && std::min(I->F.W, I->F.H) - 2 > 3;
}

int mg_solve(Instance *I)
{
    const sc_solver_opts &o = I->opts;
    int rc = build_levels(I);
    if (rc) return rc;
    {
        const int nb = std::max(std::max(prolong_blocks(I->F.W - 2, I->F.H - 2, I->F.C), tb_blocks_level0(I->F.W, I->F.H, I->F.C, 2)),
                                cycle0_blocks(I->F.W, I->F.H, I->F.C, 4));      // the deepest forms have the most workgroups
        if ((rc = ensure(I, I->mg_partial, sizeof(float) * (2 * (size_t)nb + 64)))) return rc;   // two cycles' worth (see the stop rule) + the saturation word behind them
    }
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    const float utol = o.update_tol > 0.f ? o.update_tol : 0.25f;
    const int budget = o.max_sweeps > 0 ? o.max_sweeps : 30;
    // level 1 in float16 or float: the two formats put a plane's ring and pads at different bytes, so a switch re-zeroes the planes
    const bool l1h = I->mg.size() >= 2 && mg_level1_half(I);
    if (I->mg.size() >= 2 && l1h != I->mg_l1_half) {
        MGLevel &L1 = I->mg[1];
        SC_HIP(I, hipMemsetAsync(L1.U.p, 0, L1.U.bytes(), I->stream));
        SC_HIP(I, hipMemsetAsync(L1.F.p, 0, L1.F.bytes(), I->stream));
        SC_HIP(I, hipMemsetAsync(L1.T.p, 0, L1.T.bytes(), I->stream));
        I->mg_l1_half = l1h;
    }
    // the level-0 scratch is the ping-pong partner of the solution; the residual field only
    // writes its interior, and both buffers carry the same ring, so it stays a valid partner
    int cyc = 0;
    bool ok = false;
    // Fused level-0 form: one launch per cycle does [prolongation +] post-smoothing of this cycle,
    // pre-smoothing of the next, residual and restriction (sc_cycle0.hip).  The first launch has no
    // correction to add.  A cycle whose result the stop rule is about to judge is launched in its
    // "final" form instead (prolongation + post-smoothing only): when the rule accepts it -- the normal
    // case for the third cycle -- nothing was computed for a cycle that never runs, and the field is
    // exactly the textbook V-cycle's; when it does not, one pre-smoothing + residual + restriction launch
    // (the form of the very first launch) catches up and the cycles continue.
    const bool fused0 = fused_level0(o) && I->mg.size() >= 2;
    if (I->f_half && !(fused0 && o.tol <= 0.f)) { I->err = "internal: float16 right-hand side on a path that needs float"; return SC_ERR_BAD_ARG; }
    if (fused0) {
        Field none{};
        const bool out_wanted = I->spec_post.armed && o.tol <= 0.f && !(o.flags & SC_FLAG_KEEP_FIELD) && pre == 2 && post == 2;
        const bool q16 = l1h && mg_field_q16(I, out_wanted) && budget > 1;      // (max_sweeps = 1: the first cycle is the judged one)
        const int nb_cap = cycle0_blocks(I->F.W, I->F.H, I->F.C, 4);   // deepest form = largest halo = most workgroups
        // The 16-bit stores check their range (sc_cycle0.hip, c0_q16_checked): one that saturates writes this solve's generation
        // word behind the partial maxima; the output launches then write nothing, the read-back of the maxima brings the word
        // along, and the clone is repeated on float fields (SC_RETRY_FLOAT_FIELD).  A NaN pattern: no maximum ever has these bits.
        // Up to 16384 workgroups (single clones, small groups) the host folds the per-workgroup maxima itself -- and the launches
        // store them (and the saturation word) STRAIGHT INTO PINNED HOST MEMORY: no read-back copy behind the judged cycle (a
        // command of its own on the critical path, ~4 us + its gap).  Larger grids fold on the device first.
        const bool host_fold = nb_cap <= 16384;
        if (host_fold && (rc = ensure_pinned(I, I->h_partial, sizeof(float) * (2 * (size_t)nb_cap + 64)))) return rc;
        float *const part_base = host_fold ? (float *)I->h_partial.p : (float *)I->mg_partial.p;
        AbortFlag sat;
        if (q16) {      // the device word sits behind the device list of maxima (the output launches test it there), its host copy behind the pinned one
            sat.p = (unsigned *)((float *)I->mg_partial.p + 2 * (size_t)nb_cap);
            sat.host = host_fold ? (unsigned *)((float *)I->h_partial.p + 2 * (size_t)nb_cap) : nullptr;
            sat.gen = 0x7fc00000u | (++I->sat_counter & 0x3fffffu);
        }
        I->sat = sat;
        // on the float16 path the pre-process stored the initial field as float16 as well (first launch only)
        if (launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, none, I->mg[0].g, pre, false, nullptr,
                          I->stream, false, I->f_half, I->u_half, false, nullptr, l1h, q16, sat, I->rag.dev) < 0) { I->err = "cycle0: unsupported depth"; return SC_ERR_BAD_ARG; }
        I->result_in_U1 = !I->result_in_U1;
        if (I->rag.dev && I->rag.ready_pending) {      // a size class: its zeroed coarse planes and tables were made on the second stream beside everything up to here
            SC_HIP(I, hipStreamWaitEvent(I->stream, I->rag.ev_ready, 0));      // (the matrices: run_tail waits for them)
            I->rag.ready_pending = false;
        }
        I->u_half = false;             // consumed: both U buffers hold float (or 16-bit fixed point: u_q16) from here on
        I->u_q16 = I->mg_q16_last = q16;
        I->info.sweep_launches += 1;
        int nb_last = 0;                   // workgroups (= partial maxima) of the previous cycle's level-0 launch
        // Output straight from the last cycle.  When the caller armed the splice (spec_post) the cycle the stop rule is about to
        // judge does not write its field: it adds the float-table node correction, clamps, truncates and leaves output BYTES
        // (planar, in the memory of the partner field; a small kernel interleaves them into the destination) -- 3 bytes less
        // written and 9 less read per pixel and channel than field + post-process.  The node correction it adds is the one of
        // the iterate BEFORE that cycle, whose cell shares the previous launch leaves behind (lowmode_early_kind: the two
        // differ by 0.001-0.003 grey levels, 0.05 in the worst case the stop rule admits).  If the rule rejects the cycle, the same cycle is launched again in the form
        // that writes the field (its input is untouched) and the solve continues as without this.
        auto stop_rule = [utol](float m, float m_prev) {
            if (m_prev > 0.f) {
                const float rho = std::min(0.5f, std::max(0.02f, m / m_prev));
                return m * rho / (1.0f - rho) <= 0.1f * utol;
            }
            return m <= utol;
        };
        // max |correction| of the cycle just launched = max over its per-workgroup maxima (at part_now; `cyc` already counts the
        // cycle), and of the cycle before when its maxima are at hand.  A few thousand maxima are folded here on the host, out of
        // the pinned buffer the launches wrote them to (host_fold); large grids (groups of clones) reduce both lists on the device
        // first (one launch) and copy three words.  m_prev < 0: unknown.  `output` (the splice or post-process of the result, or nothing) is enqueued between the launch
        // and the read-back: it then starts without a gap while the host waits.
        bool saturated = false;            // set by correction_maxima: a 16-bit store of this solve left its range
        auto correction_maxima = [&](int nb, int nb_prev, int nb_cap, float *part_now, const std::function<int()> &output, float &m, float &m_prev) -> int {
            m = 0.f; m_prev = -1.f;
            int orc;
            if (host_fold) {
                const bool have_prev = nb_prev > 0;                       // the launch of the previous cycle wrote the other half
                if ((orc = output())) return orc;
                SC_HIP(I, hipStreamSynchronize(I->stream));               // the maxima are in the pinned buffer when the launch has ended
                if (sat.p) { unsigned w; memcpy(&w, (const float *)I->h_partial.p + 2 * (size_t)nb_cap, sizeof(w)); saturated = w == sat.gen; }
                const float *hp = (const float *)I->h_partial.p + (size_t)(cyc & 1) * nb_cap;            // this cycle's half
                const float *hq = (const float *)I->h_partial.p + (size_t)((cyc + 1) & 1) * nb_cap;      // the previous cycle's
                for (int i = 0; i < nb; ++i) m = hp[i] > m ? hp[i] : m;
                if (have_prev) {
                    m_prev = 0.f;
                    for (int i = 0; i < nb_prev; ++i) m_prev = hq[i] > m_prev ? hq[i] : m_prev;
                }
            } else {
                const float *part_prev = part_base + (size_t)((cyc + 1) & 1) * nb_cap;
                launch_max_final2(part_now, nb, part_prev, nb_prev > 0 ? nb_prev : 0, I->d_maxcorr, I->stream, sat.p);
                if ((orc = output())) return orc;
                SC_HIP(I, hipMemcpyAsync(I->h_maxcorr, I->d_maxcorr, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, I->stream));
                SC_HIP(I, hipStreamSynchronize(I->stream));
                memcpy(&m, &I->h_maxcorr[0], sizeof(float));
                memcpy(&m_prev, &I->h_maxcorr[1], sizeof(float));
                saturated = sat.p && I->h_maxcorr[2] == sat.gen;
            }
            return SC_OK;
        };
        bool early_ready = false;          // the node correction for the judged cycle's output is on its way (early_lm; CN == nullptr: none to add)
        bool early_cond = false;           // ... and may be used only if the judged update turns out small enough (lowmode_early_kind 3)
        LmNodes early_lm;
        while (cyc < budget) {
            const bool comp1 = mg_composes_level1(I);
            if ((rc = vcycle(I, 1, pre, post, comp1 ? (1u << 1) : 0u))) return rc;
            // The first two corrections of a solve are never below the stop threshold unless the
            // initial guess was already the answer, and every check costs a host round trip
            // (~25 us), so checking starts with the third cycle.
            const bool judged = !(cyc + 1 < 3 && cyc + 1 < budget && o.tol <= 0.f);
            float *const part_now = part_base + (size_t)((cyc + 1) & 1) * nb_cap;    // this cycle's maxima; the previous cycle's sit in the other half
            // the judged cycle runs in its final form; when the float-table correction will follow it leaves the correction's
            // cell shares behind (sc_lowmode.hip), which saves the correction its own pass over the field
            const bool next_judged = !judged && !(cyc + 2 < 3 && cyc + 2 < budget && o.tol <= 0.f);
            // lowmode_early_kind 3: the a-priori bound does not cover this size (the float tables' low modes are off by more than 4 %);
            // the output may still carry the earlier iterate's correction IF the judged cycle's measured update keeps the difference
            // below the same 0.049 grey levels -- decided with the stop rule, below (early_cond)
            const int early_kind = (out_wanted && next_judged) ? lowmode_early_kind(I, utol) : 2;
            const int early = early_kind == 3 ? 1 : early_kind;
            if (out_wanted && next_judged) early_cond = early_kind == 3;
            float4 *const bands = legacy_path(o, SC_LEGACY_SEPARATE_RESTRICT) ? nullptr
                                  : judged ? lowmode_bands_buffer(I, post) : early == 1 ? lowmode_bands_buffer(I, post + pre) : nullptr;
            if (judged && early_ready) {
                if (I->aux_pending) {          // the node correction is ready when the launch that adds it starts
                    SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_join, 0));
                    I->aux_pending = false;
                }
                const int nbo = launch_cycle0_out(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U, I->mg[0].g, part_now,
                                                  I->stream, I->f_half, comp1, comp1 ? I->mg[2].U : Field(), comp1 ? I->mg[1].g : MGGeom(), early_lm, l1h, I->rag.dev);
                early_ready = false;
                if (nbo > 0) {
                    I->info.sweep_launches += 1;
                    ++cyc;
                    SC_HIP(I, hipGetLastError());
                    const Field Q = I->result_in_U1 ? I->U0 : I->U1;
                    float m, m_prev;
                    if ((rc = correction_maxima(nbo, nb_last, nb_cap, part_now, [&]() -> int {
                            if (I->spec_post.group.empty()) launch_splice_planar(Q, I->spec_post.body_org, I->spec_post.bstep, I->stream, I->guard, sat);
                            else launch_splice_planar_group(Q, I->spec_post.group.data(), (int)I->spec_post.group.size(), I->stream, sat);
                            return SC_OK;
                        }, m, m_prev))) return rc;
                    I->info.last_update = m;
                    if (saturated) { I->info.sweeps = cyc; return SC_RETRY_FLOAT_FIELD; }      // nothing was written (AbortFlag)
                    if (stop_rule(m, m_prev) && !(early_cond && I->lm.max_ratio * (double)m > 0.049)) { I->spec_post.done = true; I->out_direct = true; ok = true; break; }
                    // rejected: the same cycle again in the form that keeps the field, then on as usual
                    --cyc;
                    I->info.sweep_launches -= 1;
                }
            }
            const int nb = comp1
                ? launch_cycle0_composed(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U, I->mg[0].g,
                                         judged ? post : post + pre, part_now, I->stream, false, I->f_half, judged,
                                         I->mg[2].U, I->mg[1].g, bands, l1h, I->u_q16 ? (next_judged ? 1 : 3) : 0, sat, I->rag.dev)
                : launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U,
                                I->mg[0].g, judged ? post : post + pre, true, part_now, I->stream,
                                false, I->f_half, false, judged, bands);
            if (nb <= 0) { I->err = "cycle0: unsupported depth"; return SC_ERR_BAD_ARG; }
            I->result_in_U1 = !I->result_in_U1;
            if (next_judged) I->u_q16 = false;      // the launch before the judged cycle left a float field
            lowmode_bands_written(I, bands ? result(I).p : nullptr);
            if (!judged && early != 2) {       // the node correction the next cycle's output will carry, from this launch's field
                early_lm = LmNodes();
                // a group of clones: its coarse levels fill the chip, nothing to overlap (measured: -2 %); a small clone: the two
                // cross-stream waits cost more than the 15-us chain they hide (154x100 ... 300x194 patches: +20 us; neutral at 730^2 ... 800^2,
                // -2..3 % from 900^2 on: the threshold is 0.79 Mpix, 1 Mpix until late in round 4)
                if (early == 1 && (I->F.C > 3 || (size_t)I->F.W * I->F.H < (size_t)3 << 18)) {
                    if ((rc = lowmode_nodes(I, result(I), early_lm))) return rc;
                } else if (early == 1) {       // one large clone: on the second stream, beside the coarse levels of the next cycle (-15 us of 500 at 2048^2)
                    SC_HIP(I, hipEventRecord(I->ev_fork, I->stream));
                    SC_HIP(I, hipStreamWaitEvent(I->aux, I->ev_fork, 0));
                    if ((rc = lowmode_nodes(I, result(I), early_lm, I->aux))) return rc;
                    SC_HIP(I, hipEventRecord(I->ev_join, I->aux));
                    I->aux_pending = true;
                }
                early_ready = true;            // (early == 0: nothing to add)
            }
            I->info.sweep_launches += 1;
            ++cyc;
            SC_HIP(I, hipGetLastError());
            const int nb_prev = nb_last;
            nb_last = nb;
            if (!judged) continue;
            // the post-process goes in FIRST (see Instance::spec_post), the read-back of the maxima follows it
            float m, m_prev;
            if ((rc = correction_maxima(nb, nb_prev, nb_cap, part_now, [&]() -> int {
                    if (!(I->spec_post.armed && o.tol <= 0.f)) return SC_OK;
                    LmNodes lm;
                    const int lrc = output_nodes(I, lm);
                    if (lrc) return lrc;
                    if (I->spec_post.group.empty()) launch_postprocess(result(I), I->spec_post.body_org, I->spec_post.bstep, I->stream, I->guard, lm, sat);
                    else launch_postprocess_group(result(I), I->spec_post.group.data(), (int)I->spec_post.group.size(), I->stream, lm, sat);
                    I->spec_post.done = true;
                    return SC_OK;
                }, m, m_prev))) return rc;
            I->info.last_update = m;
            if (saturated) { I->spec_post.done = false; I->info.sweeps = cyc; return SC_RETRY_FLOAT_FIELD; }
            if (o.tol > 0.f) {
                double r[2];
                if ((rc = eval_residual(I, r))) return rc;
                const double rel = (r[1] > 0.0) ? std::sqrt(r[0] / r[1]) : std::sqrt(r[0]);
                I->info.rel_residual = rel;
                if (rel <= (double)o.tol) { ok = true; break; }
            }
            // Stop rule.  The error left after a cycle is about rho / (1 - rho) times the correction it applied, rho being the
            // contraction per cycle (measured: the prediction matches the next correction to ~10 %).  With two successive
            // corrections known the bound is applied to that prediction: error <= 0.1 x update_tol (0.025 grey levels at the
            // default 0.25, i.e. the error the plain threshold "correction <= update_tol" leaves at rho = 0.09).  A solve that
            // contracts faster stops on a larger last correction, a slower one on a smaller.  Without a previous correction
            // (max_sweeps = 1) the plain threshold decides.
            if (stop_rule(m, m_prev)) { ok = true; break; }
            I->spec_post.done = false;     // not converged: the field moves on, the output is written again later
            if (cyc < budget) {            // catch up: pre-smoothing + residual + restriction for the next cycle
                if (launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, none, I->mg[0].g, pre, false,
                                  nullptr, I->stream, false, I->f_half, false, false, nullptr, l1h, false, AbortFlag(), I->rag.dev) < 0) { I->err = "cycle0: unsupported depth"; return SC_ERR_BAD_ARG; }
                I->result_in_U1 = !I->result_in_U1;
                lowmode_bands_written(I, nullptr);
                I->info.sweep_launches += 1;
            }
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

// Measurement hook (sc_hip_time_coarse_chain): the launch-bound part of a cycle -- levels 2 .. bottom .. 2: seven dependent launches
// for a 2048^2 clone, 2 % of the unknowns -- run `reps` times back to back on the hierarchy the last multigrid solve left, (a) as
// plain launches and (b) captured once into a HIP graph and replayed.  hipEvents on the instance's stream around each batch.
// Values are discarded (level 2's right-hand side is whatever the last cycle left there).
int mg_time_coarse_chain(Instance *I, int reps, float *ms_eager, float *ms_graph, int *launches)
{
    if (I->mg.size() < 4 || I->mg_bottom < 3 || !mg_composes_level1(I)) { I->err = "time_coarse_chain: run a multigrid clone of at least ~500^2 first"; return SC_ERR_BAD_ARG; }
    const sc_solver_opts &o = I->opts;
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    int rc;
    if (I->fd_pending) { SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_fd, 0)); I->fd_pending = false; }
    *launches = (int)(2 * (I->mg_bottom - 2) + 1) - (tail_serves(I, I->mg_bottom - 1) ? 2 : 0);
    if ((rc = vcycle(I, 2, pre, post))) return rc;                       // warm
    SC_HIP(I, hipEventRecord(I->ev_k0, I->stream));
    for (int i = 0; i < reps; ++i) if ((rc = vcycle(I, 2, pre, post))) return rc;
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    float ms = 0.f;
    SC_HIP(I, hipEventElapsedTime(&ms, I->ev_k0, I->ev_k1));
    *ms_eager = ms / (float)reps;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    SC_HIP(I, hipStreamBeginCapture(I->stream, hipStreamCaptureModeThreadLocal));
    rc = vcycle(I, 2, pre, post);
    hipError_t e = hipStreamEndCapture(I->stream, &graph);
    if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); return rc ? rc : hip_fail(I, e, "hipStreamEndCapture"); }
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); return hip_fail(I, e, "hipGraphInstantiate"); }
    (void)hipGraphLaunch(exec, I->stream);                                 // warm (uploads the executable graph)
    SC_HIP(I, hipEventRecord(I->ev_k0, I->stream));
    for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(exec, I->stream);
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    e = hipStreamSynchronize(I->stream);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hip_fail(I, e, "hipStreamSynchronize");
    SC_HIP(I, hipEventElapsedTime(&ms, I->ev_k0, I->ev_k1));
    *ms_graph = ms / (float)reps;
    return SC_OK;
}

// Measurement hook (sc_hip_time_tail_phases): one k_mg_tail launch on the hierarchy the last multigrid solve left, with the shader
// clock of channel 0's first thread at its eleven phase boundaries: entry | right-hand side in registers | pre-smoothing done |
// residual + restriction done (level B's right-hand side in LDS) | products 1, 2, 3, 4 | prolongation + edge exchange |
// post-smoothing | stores issued.  Differences are cycles of the shader clock.
int mg_time_tail_phases(Instance *I, unsigned long long *out11)
{
    if (I->mg.size() < 4 || I->mg_bottom < 3 || !tail_serves(I, I->mg_bottom - 1)) { I->err = "time_tail_phases: the last run was not a multigrid solve whose bottom runs as k_mg_tail"; return SC_ERR_BAD_ARG; }
    const sc_solver_opts &o = I->opts;
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    unsigned long long *d = nullptr;
    SC_HIP(I, hipMalloc(&d, 11 * sizeof(unsigned long long)));
    bool done = false;
    int rc = run_tail(I, I->mg_bottom - 1, pre, post, done);             // warm
    if (!rc) rc = run_tail(I, I->mg_bottom - 1, pre, post, done, d);
    hipError_t e = hipStreamSynchronize(I->stream);
    if (!rc && e == hipSuccess && done) e = hipMemcpy(out11, d, 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(I, e, "time_tail_phases");
    return done ? SC_OK : SC_ERR_BAD_ARG;
}

} // namespace sc
