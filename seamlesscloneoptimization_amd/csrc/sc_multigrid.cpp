// sc_multigrid.cpp -- multigrid V-cycle driver (placeholder: SOR to tolerance).
#include "sc_instance.h"
#include <cmath>
namespace sc {
int mg_solve(Instance *I)
{
    const sc_solver_opts &o = I->opts;
    const int budget = 200000, every = 64;
    int done = 0;
    while (done < budget) {
        int rc = run_sweeps(I, SC_METHOD_SOR, every, 0.f, 1);
        if (rc) return rc;
        done += every;
        double r[2];
        if ((rc = eval_residual(I, r))) return rc;
        const double rel = (r[1] > 0.0) ? std::sqrt(r[0] / r[1]) : std::sqrt(r[0]);
        I->info.rel_residual = rel;
        I->info.sweeps = done;
        if (rel <= (double)(o.tol > 0.f ? o.tol : 1e-6f)) { I->info.converged = 1; return SC_OK; }
    }
    return SC_ERR_NOT_CONVERGED;
}
} // namespace sc
