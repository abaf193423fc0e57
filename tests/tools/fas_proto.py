"""Prototype: band-adaptive FAS multigrid (MLAT) for the seamless-clone system, numpy float64.
u0 = (eroded mask ? patch : dst); e = u - u0 solves A e = r0 (zero Dirichlet), r0 = f - A u0 is exactly zero away from
the mask's edges.  Levels 0 and 1 exist only in a band around the driven pixels; level 2 is dense and solved exactly here."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import oracle_np as o, mg_np as mg

def coefs(dx, dy):
    cw, ddx, cn, ddy = (a.astype(np.float64) for a in mg._coefs(dx, dy))
    return cw, ddx, cn, ddy

def apply_A(U, dx, dy):
    cw, ddx, cn, ddy = coefs(dx, dy)
    out = np.zeros_like(U)
    out[1:-1, 1:-1] = ((cw[None, 1:-1] * U[1:-1, :-2] + U[1:-1, 2:]) + (cn[1:-1, None] * U[:-2, 1:-1] + U[2:, 1:-1])
                       - (ddx[None, 1:-1] + ddy[1:-1, None]) * U[1:-1, 1:-1])
    return out

def smooth(U, F, dx, dy, region, sweeps):
    cw, ddx, cn, ddy = coefs(dx, dy)
    H, W = U.shape
    yy, xx = np.mgrid[0:H, 0:W]
    U = U.copy()
    for _ in range(sweeps):
        for color in (0, 1):
            m = region & (((xx + yy) & 1) == color)
            s = np.zeros_like(U)
            s[1:-1, 1:-1] = ((cw[None, 1:-1] * U[1:-1, :-2] + U[1:-1, 2:]) + (cn[1:-1, None] * U[:-2, 1:-1] + U[2:, 1:-1]))
            gs = np.zeros_like(U)
            gs[1:-1, 1:-1] = (s[1:-1, 1:-1] - F[1:-1, 1:-1]) / (ddx[None, 1:-1] + ddy[1:-1, None])
            U[m] = gs[m]
    return U

def inject(U, dxc, dyc):
    out = np.zeros((dyc.n + 2, dxc.n + 2))
    out[1:-1, 1:-1] = U[2:2 * dyc.n + 1:2, 2:2 * dxc.n + 1:2]
    return out

def region_frame(dx, dy, d):
    H, W = dy.n + 2, dx.n + 2
    yy, xx = np.mgrid[0:H, 0:W]
    dist = np.minimum(np.minimum(xx, W - 1 - xx), np.minimum(yy, H - 1 - yy))
    r = dist <= d
    r[0, :] = r[-1, :] = False; r[:, 0] = r[:, -1] = False
    return r

def fas(levels, r0, regions, dense_level, cycles=3, nu=(2, 2), log=None, exact=None):
    # full approximations per level (persist)
    u = [np.zeros((dy.n + 2, dx.n + 2)) for dx, dy in levels[:dense_level + 1]]
    f = [None] * (dense_level + 1)
    f[0] = r0
    def cyc(l):
        dx, dy = levels[l]
        if l == dense_level:
            u[l] = mg.solve_exact(f[l].astype(np.float32), dx, dy).astype(np.float64) if False else solve_dense(f[l], dx, dy)
            return
        reg = regions[l]
        u[l] = smooth(u[l], f[l], dx, dy, reg, nu[0])
        r = np.where(reg, f[l] - apply_A(u[l], dx, dy), 0.0)
        dxc, dyc = levels[l + 1]
        under = inject(reg.astype(np.float64), dxc, dyc) > 0           # coarse points that have a fine counterpart in the region
        uc_init = np.where(under, inject(u[l], dxc, dyc), u[l + 1])
        fc = mg.restrict(r.astype(np.float32), dx, dy).astype(np.float64) if False else restrict64(r, dx, dy)
        f[l + 1] = np.where(under, fc + apply_A(uc_init, dxc, dyc), fc + 0.0 * fc if l + 1 < dense_level and False else np.where(under, 0, f_orig(l + 1)))
        # careful: outside 'under' the coarse equation is the original one (zero RHS in the e-formulation), but the restriction of
        # residuals from fine points near the region edge also reaches coarse points just outside 'under': add it there
        f[l + 1] = np.where(under, fc + apply_A(uc_init, dxc, dyc), fc)
        u[l + 1] = uc_init
        cyc(l + 1)
        corr = mg.prolong((u[l + 1] - uc_init).astype(np.float32), dx, dy).astype(np.float64) if False else prolong64(u[l + 1] - uc_init, dx, dy)
        full = prolong64(u[l + 1], dx, dy)
        u[l] = np.where(reg, u[l] + corr, full)
        u[l] = smooth(u[l], f[l], dx, dy, reg, nu[1])
    for c in range(cycles):
        cyc(0)
        if exact is not None:
            err = np.abs(u[0] - exact)
            print(f"  cycle {c + 1}: max err {err.max():.4f}  in band {err[regions[0]].max():.4f}  outside {err[~regions[0]].max():.4f}", flush=True)
    return u[0]

def f_orig(l): return 0.0

def interp_mats(dx, dy):
    return mg.interp_matrix(dx), mg.interp_matrix(dy)

_cache = {}
def prolong64(E, dx, dy):
    key = ('p', dx.n, dx.alpha, dy.n, dy.alpha)
    if key not in _cache: _cache[key] = (mg.interp_matrix(dy), mg.interp_matrix(dx))
    Py, Px = _cache[key]
    return Py @ E @ Px.T

def restrict64(R, dx, dy):
    key = ('r', dx.n, dx.alpha, dy.n, dy.alpha)
    if key not in _cache:
        Py, Px = mg.interp_matrix(dy), mg.interp_matrix(dx)
        sy = Py.T.sum(1, keepdims=True); sy[sy == 0] = 1; sx = Px.T.sum(1, keepdims=True); sx[sx == 0] = 1; Ry = Py.T / sy; Rx = Px.T / sx
        _cache[key] = (Ry, Rx)
    Ry, Rx = _cache[key]
    out = 4.0 * (Ry @ R @ Rx.T)
    out[0, :] = out[-1, :] = 0; out[:, 0] = out[:, -1] = 0
    return out

def solve_dense(F, dx, dy):
    import scipy.sparse as sp, scipy.sparse.linalg as spl
    cw, ddx, cn, ddy = coefs(dx, dy)
    nx, ny = dx.n, dy.n
    def T(n, cwv, dd):
        main = -dd[1:n + 1]; lower = cwv[2:n + 1]; upper = np.ones(n - 1)
        return sp.diags([lower, main, upper], [-1, 0, 1], shape=(n, n))
    A = sp.kron(sp.identity(ny), T(nx, cw, ddx)) + sp.kron(T(ny, cn, ddy), sp.identity(nx))
    x = spl.spsolve(A.tocsc(), F[1:-1, 1:-1].ravel())
    U = np.zeros_like(F); U[1:-1, 1:-1] = x.reshape(ny, nx)
    return U

if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    H = int(sys.argv[2]) if len(sys.argv) > 2 else W
    d0 = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    d1 = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    dense = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=32)
    geo = o.mask_stage(mask, cx, cy)
    B, lap, g = o.build_rhs(dst, patch, geo, dtype=np.float64)
    c = 1
    uex = o.full_field(B, o.solve_dst(g))[:, :, c]
    M = geo["M"]; P = patch[geo["y0"]:geo["y0"] + H, geo["x0"]:geo["x0"] + W, c].astype(np.float64)
    u0 = np.where(M == 255, P, B[:, :, c])
    levels = mg.build_levels(W, H)
    dx, dy = levels[0]
    r0 = lap[:, :, c] - apply_A(u0, dx, dy)
    r0[0, :] = r0[-1, :] = 0; r0[:, 0] = r0[:, -1] = 0
    print("nonzero r0 rows/cols distance from ring:", np.max(np.nonzero(np.abs(r0).sum(1) > 0)[0][:10]), "max |r0|", np.abs(r0).max(), "levels", [(a.n, round(a.alpha, 3)) for a, _ in levels[:4]])
    eex = uex - u0
    print("exact e: max", np.abs(eex).max())
    regions = [region_frame(*levels[0], d0), region_frame(*levels[1], d1)] + [region_frame(*levels[l], 10 ** 9) for l in range(2, dense + 1)]
    for l in range(dense): print(f"level {l}: region {regions[l].mean() * 100:.1f} % of {regions[l].size} points")
    t = time.time()
    e = fas(levels, r0, regions, dense, cycles=4, exact=eex)
    print("time", round(time.time() - t, 1), "s")
