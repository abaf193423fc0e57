// sc_solver.cpp -- host-side drivers of the iterative Poisson solve:
// fixed-count or residual-terminated Jacobi / red-black GS / SOR sweeps, and the multigrid
// V-cycle built from the same smoothers.  The reference has no iterative solver (it inverts
// the same 5-point system with a DST, seamlessClone_imp.cpp:1814-1896); these drivers
// converge to that system's solution.
#include "sc_instance.h"
#include <algorithm>

namespace sc {

Field &result(Instance *I) { return I->result_in_U1 ? I->U1 : I->U0; }
static Field &other(Instance *I) { return I->result_in_U1 ? I->U0 : I->U1; }

// What the post-process adds to result(I).  For the converged multigrid solve that is, by default, the float-table
// correction (sc_lowmode.hip: result + correction = the answer OpenCV and the reference compute); the solution itself
// stays untouched (the solve may continue if the stop rule rejects the cycle, and the diagnostic hooks read it).
// SC_FLAG_EXACT_TABLES, the sweep solvers (fixed counts, not converged fields) and ROIs without unknowns: nothing.
// SC_METHOD_AUTO: the direct solve where it costs no more than the cycles (small ROIs: a handful of launches either way, and
// the direct form has no iteration error), multigrid above; a residual-based stop (tol > 0) is an iterative notion.
int effective_method(const Instance *I)
{
    const sc_solver_opts &o = I->opts;
    if (o.method != SC_METHOD_AUTO) return o.method;
    const int w = I->F.W - 2, h = I->F.H - 2;
    if (o.tol > 0.f || w < 1 || h < 1) return SC_METHOD_MULTIGRID;
    // a GROUP of clones (3n channels through one set of launches) is about throughput, and there the cycles win at every size:
    // 7.5 / 12.1 / 15.1 Gpix/s against 2.1 / 3.3 / 3.5 for the double-precision direct solve at 300^2 / 512^2 / 768^2 ROIs in groups of 16
    if (I->F.C > 3) return SC_METHOD_MULTIGRID;
    // the direct solve = the FFT form with double transforms (fft_in_double): the answer of the matrix form SC_METHOD_DST (both
    // are the reference's float-table arithmetic with exact transforms; measured diff sums against the port are identical) in
    // 0.09-0.29 ms of device time where the matrix form takes 0.17-0.37 and the cycles 0.18-0.35 (298x192 ... 896^2 ROIs; at 1024^2 the
    // cycles win again: 0.295 against 0.313 ms -- tools/fft_probe.py; round 4, tools/size_probe.py [--mg]: the cycles win from ~730^2 on:
    // 0.221 against 0.212 ms at 700^2, 0.222 against 0.231 at 750^2, 0.228 against 0.268 at 900^2)
    if (w <= SC_AUTO_DIRECT_MAX && h <= SC_AUTO_DIRECT_MAX) return SC_METHOD_FFT;
    // elongated ROIs (round 4, tools/size_probe.py --fft / --mg): the transforms along the short side are short and few rows make the
    // long ones -- 900 x 100: 0.101 ms directly against 0.191 in cycles, 1000 x 400: 0.164 / 0.199, 1500 x 200: 0.191 / 0.223,
    // 2048 x 100: 0.172 / 0.222, 4000 x 130: 0.371 / 0.445 (130 x 4000: 0.345 / 0.559), 4090 x 60: 0.255 / 0.301; the cycles keep
    // 1000 x 700 (0.250 / 0.237), 1200 x 600 (0.350 / 0.252), 2000 x 500 (0.335 / 0.278), 3000 x 300 (0.533 / 0.300)
    if (std::max(w, h) <= SC_AUTO_THIN_LONG_MAX &&
        ((long)w * h <= SC_AUTO_DIRECT_AREA || std::min(w, h) <= SC_AUTO_NARROW_MAX)) return SC_METHOD_FFT;
    // ROIs narrower than 7 pixels (the three erodes empty the mask: the exact solution is the destination itself, integers):
    // the reference's float tables put every mode ~1e-7 below its exact value, so its answer is v - epsilon and truncates to v - 1
    // almost everywhere.  Only the direct form reproduces that (it IS that arithmetic); it stays cheap while one side is tiny.
    if (std::min(w, h) <= SC_AUTO_THIN_MAX && std::max(w, h) <= SC_AUTO_THIN_LONG_MAX) return SC_METHOD_FFT;
    return SC_METHOD_MULTIGRID;
}

// SC_METHOD_FFT: transforms in double when the caller asks (SC_FLAG_FFT_FP64) and always when SC_METHOD_AUTO chose it
bool fft_in_double(const Instance *I)
{
    return (I->opts.flags & SC_FLAG_FFT_FP64) || I->opts.method == SC_METHOD_AUTO;
}

bool wants_float_tables(const Instance *I)
{
    return effective_method(I) == SC_METHOD_MULTIGRID && !(I->opts.flags & SC_FLAG_EXACT_TABLES) && I->F.W >= 3 && I->F.H >= 3;
}

int output_nodes(Instance *I, LmNodes &lm)
{
    lm = LmNodes();
    if (!wants_float_tables(I)) return SC_OK;
    return lowmode_nodes(I, result(I), lm);
}

float optimal_omega(int W, int H)
{
    const double w = W - 2, h = H - 2;
    if (w < 1 || h < 1) return 1.0f;
    const double rho = 0.5 * (std::cos(M_PI / (w + 1.0)) + std::cos(M_PI / (h + 1.0)));
    return (float)(2.0 / (1.0 + std::sqrt(std::max(0.0, 1.0 - rho * rho))));
}

int eval_residual(Instance *I, double out[2])
{
    launch_residual(result(I), I->F, I->d_partials, I->d_red, I->stream);
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipMemcpyAsync(I->h_red, I->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    out[0] = I->h_red[0];
    out[1] = I->h_red[1];
    return SC_OK;
}

// sweeps_per_launch semantics (opts / hooks):
//    0  library default: fused register-blocked kernels at their deepest supported depth
//    1  one sweep per launch with the plain kernels (k_jacobi; two k_rb_half launches)
//   -1  fused kernel, depth 1 (one full red-black sweep = both colours in one launch)
//  >=2  fused kernel, that depth (clamped to what the kernel supports)
// All variants produce bit-identical fields; a remainder < depth runs through a shallower launch.
int fused_depth(int method, int spl)
{
    if (spl == 1) return 0;             // plain kernels
    if (spl == 0) return tb_max_depth(method);
    if (spl < 0) return 1;
    const int mx = tb_hard_max_depth(method);
    return spl < mx ? spl : mx;
}

int run_sweeps(Instance *I, int method, int sweeps, float omega, int spl)
{
    field_moved(I);
    if (sweeps <= 0) return SC_OK;
    if (method != SC_METHOD_JACOBI && method != SC_METHOD_RBGS && method != SC_METHOD_SOR) {
        I->err = "run_sweeps: unknown method";
        return SC_ERR_BAD_ARG;
    }
    const int depth = fused_depth(method, spl);
    float om = 1.0f;
    if (method == SC_METHOD_SOR) om = (omega > 0.f) ? omega : optimal_omega(I->F.W, I->F.H);
    int left = sweeps;
    while (left > 0) {
        int T = 1;
        bool done = false;
        // deepest instantiated depth <= min(depth, left) (the launchers return false, without
        // launching, for a depth that is not instantiated)
        for (int t = std::min(depth, left); t >= 1 && !done; --t) {
            done = (method == SC_METHOD_JACOBI) ? launch_jacobi_tb(result(I), other(I), I->F, t, I->stream, I->bench_tag)
                                                : launch_rb_tb(result(I), other(I), I->F, t, om, I->stream, I->bench_tag);
            if (done) { T = t; I->result_in_U1 = !I->result_in_U1; I->info.sweep_launches += 1; }
        }
        if (!done) {
            T = 1;
            if (method == SC_METHOD_JACOBI) {
                launch_jacobi(result(I), other(I), I->F, I->stream, I->bench_tag, I->opts.jacobi_tile_rows);
                I->result_in_U1 = !I->result_in_U1;
                I->info.sweep_launches += 1;
            } else {
                launch_rb_half(result(I), I->F, 0, om, I->stream, I->bench_tag);
                launch_rb_half(result(I), I->F, 1, om, I->stream, I->bench_tag);
                I->info.sweep_launches += 2;
            }
        }
        left -= T;
    }
    SC_HIP(I, hipGetLastError());
    return SC_OK;
}

int mg_solve(Instance *I); // sc_multigrid.cpp

int solve(Instance *I)
{
    const sc_solver_opts &o = I->opts;
    field_moved(I);
    if (I->aux_pending) {      // a solve that ended early left work on the second stream: order it before anything new
        SC_HIP(I, hipStreamWaitEvent(I->stream, I->ev_join, 0));
        I->aux_pending = false;
    }
    I->out_direct = false;
    I->info.sweeps = 0;
    I->info.converged = 0;
    I->info.rel_residual = NAN;
    I->info.method = o.method;
    if (I->F.W < 3 || I->F.H < 3) { I->info.converged = 1; return SC_OK; } // no unknowns
    const int method = effective_method(I);
    I->info.method = method;
    if (method == SC_METHOD_MULTIGRID) return mg_solve(I);
    if (method == SC_METHOD_DST) return dst_solve(I);
    if (method == SC_METHOD_FFT) return fft_solve(I, fft_in_double(I));
    if (o.tol <= 0.f) {
        int rc = run_sweeps(I, o.method, o.max_sweeps, o.omega, o.sweeps_per_launch);
        if (rc) return rc;
        I->info.sweeps = o.max_sweeps;
        I->info.converged = 1;
        return SC_OK;
    }
    const int every = std::max(1, o.check_every);
    int done = 0;
    while (done < o.max_sweeps) {
        const int n = std::min(every, o.max_sweeps - done);
        int rc = run_sweeps(I, o.method, n, o.omega, o.sweeps_per_launch);
        if (rc) return rc;
        done += n;
        double r[2];
        rc = eval_residual(I, r);
        if (rc) return rc;
        const double rel = (r[1] > 0.0) ? std::sqrt(r[0] / r[1]) : std::sqrt(r[0]);
        I->info.rel_residual = rel;
        I->info.sweeps = done;
        if (rel <= (double)o.tol) { I->info.converged = 1; return SC_OK; }
    }
    I->info.sweeps = done;
    return SC_ERR_NOT_CONVERGED;
}

} // namespace sc
