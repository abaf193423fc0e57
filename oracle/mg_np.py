"""numpy restatement of the multigrid components the HIP library uses (TEST INFRASTRUCTURE ONLY).

The reference has no multigrid (it solves directly, seamlessClone_imp.cpp:1814-1896); this file
is the specification the GPU multigrid kernels are checked against: level geometry with one
irregular last interval per direction, general red-black smoother, residual in float64,
row-normalised transposed-interpolation restriction, bilinear prolongation.  The fixed point of
the cycle is the solution of the same 5-point system oracle_np.solve_dst inverts, which is what
the end-to-end parity tests compare against.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def coarsen_1d(n: int, a: float):
    if n % 2 == 1:
        return (n - 1) // 2, (1.0 + a) / 2.0
    if a >= 1.0:
        return n // 2, a / 2.0
    return n // 2 - 1, 1.0 + a / 2.0


class Dim:
    def __init__(self, n, a, nc):
        self.n, self.alpha, self.nc = n, float(a), nc
        self.cw_last = F32(2.0 / (1.0 + a))
        self.d_last = F32(2.0 / a)
        tail = n - 2 * nc
        D = n + a - 2.0 * nc
        self.tw1 = F32(1.0 - 1.0 / D) if tail >= 1 else F32(0)
        self.tw2 = F32(1.0 - 2.0 / D) if tail >= 2 else F32(0)
        self.inv_last = F32(1.0 / (1.5 + float(self.tw1) + float(self.tw2)))


def build_levels(W: int, H: int):
    """[(Dim x, Dim y)] from the ROI size, level 0 first (stop when min(n) <= 3)."""
    ls = [(W - 2, 1.0, H - 2, 1.0)]
    while min(ls[-1][0], ls[-1][2]) > 3:
        nx, ax = coarsen_1d(ls[-1][0], ls[-1][1])
        ny, ay = coarsen_1d(ls[-1][2], ls[-1][3])
        if nx < 1 or ny < 1:
            break
        ls.append((nx, ax, ny, ay))
    out = []
    for i, (nx, ax, ny, ay) in enumerate(ls):
        ncx = ls[i + 1][0] if i + 1 < len(ls) else 0
        ncy = ls[i + 1][2] if i + 1 < len(ls) else 0
        out.append((Dim(nx, ax, ncx), Dim(ny, ay, ncy)))
    return out


def _coefs(dx: Dim, dy: Dim):
    W, H = dx.n + 2, dy.n + 2
    cw = np.ones(W, F32); ddx = np.full(W, 2, F32)
    cw[dx.n] = dx.cw_last; ddx[dx.n] = dx.d_last
    cn = np.ones(H, F32); ddy = np.full(H, 2, F32)
    cn[dy.n] = dy.cw_last; ddy[dy.n] = dy.d_last
    return cw, ddx, cn, ddy


def rb_gen(U, F, dx: Dim, dy: Dim, sweeps=1, omega=1.0):
    """General red-black sweep on one plane (ring = 0), float32, kernel operation order:
    gs = (((cw*l + r) + (cn*u + d)) - f) / (dx + dy)."""
    U = U.astype(F32, copy=True)
    H, W = U.shape
    cw, ddx, cn, ddy = _coefs(dx, dy)
    yy, xx = np.mgrid[0:H, 0:W]
    inter = np.zeros((H, W), bool); inter[1:-1, 1:-1] = True
    for _ in range(sweeps):
        for color in (0, 1):
            m = inter & (((xx + yy) & 1) == color)
            s = np.zeros_like(U)
            s[1:-1, 1:-1] = ((cw[None, 1:-1] * U[1:-1, :-2] + U[1:-1, 2:])
                             + (cn[1:-1, None] * U[:-2, 1:-1] + U[2:, 1:-1]))
            gs = np.zeros_like(U)
            gs[1:-1, 1:-1] = (s[1:-1, 1:-1] - F[1:-1, 1:-1]) / (ddx[None, 1:-1] + ddy[1:-1, None])
            new = gs if omega == 1.0 else U + F32(omega) * (gs - U)
            U[m] = new[m]
    return U


def residual_field(U, F, dx: Dim, dy: Dim):
    """R = F - A U on the interior, float64 arithmetic from float32 values, stored float32."""
    cw, ddx, cn, ddy = (a.astype(np.float64) for a in _coefs(dx, dy))
    U64 = U.astype(np.float64)
    R = np.zeros_like(U64)
    s = ((cw[None, 1:-1] * U64[1:-1, :-2] + U64[1:-1, 2:]) + (cn[1:-1, None] * U64[:-2, 1:-1] + U64[2:, 1:-1])
         - (ddx[None, 1:-1] + ddy[1:-1, None]) * U64[1:-1, 1:-1])
    R[1:-1, 1:-1] = F[1:-1, 1:-1].astype(np.float64) - s
    return R.astype(F32)


def interp_matrix(d: Dim):
    """(n+2) x (nc+2) 1-D interpolation incl. ring rows/cols."""
    P = np.zeros((d.n + 2, d.nc + 2), np.float64)
    for i in range(1, d.n + 1):
        if i <= 2 * d.nc:
            if i % 2 == 0:
                P[i, i // 2] = 1.0
            else:
                P[i, (i - 1) // 2] += 0.5
                P[i, (i + 1) // 2] += 0.5
        else:
            P[i, d.nc] = float(d.tw1) if i - 2 * d.nc == 1 else float(d.tw2)
    P[:, 0] = 0
    P[:, d.nc + 1] = 0
    return P


def restrict(R, dx: Dim, dy: Dim):
    """Fc = 4 * (row-normalised P^T) R, interior of the coarse plane."""
    Px, Py = interp_matrix(dx), interp_matrix(dy)
    Rx = Px.T.copy(); sx = Rx.sum(1); sx[sx == 0] = 1; Rx /= sx[:, None]
    Ry = Py.T.copy(); sy = Ry.sum(1); sy[sy == 0] = 1; Ry /= sy[:, None]
    Fc = 4.0 * (Ry @ R.astype(np.float64) @ Rx.T)
    Fc[0, :] = Fc[-1, :] = 0
    Fc[:, 0] = Fc[:, -1] = 0
    return Fc.astype(F32)


def prolong(E, dx: Dim, dy: Dim):
    """P E on the fine plane (ring rows/cols = 0)."""
    Px, Py = interp_matrix(dx), interp_matrix(dy)
    out = Py @ E.astype(np.float64) @ Px.T
    out[0, :] = out[-1, :] = 0
    out[:, 0] = out[:, -1] = 0
    return out.astype(F32)


# ---- which level the library's bottom kernel solves directly (mirrors sc_multigrid.cpp: bottom_start, build_fd)
MG_BOTTOM_LDS_BYTES = 152 * 1024
MG_BOTTOM_MAX_LEVELS = 12


def _bottom_floats(dx: Dim, dy: Dim) -> int:
    return 2 * ((dx.n + 2) | 1) * (dy.n + 2)


def bottom_start(levels) -> int:
    for l in range(1, len(levels)):
        if len(levels) - l > MG_BOTTOM_MAX_LEVELS:
            continue
        if 4 * sum(_bottom_floats(*levels[k]) for k in range(l, len(levels))) <= MG_BOTTOM_LDS_BYTES:
            return l
    return len(levels)


def bottom_level(levels, matrix_cores=True):
    """The library's I->mg_bottom (sc_multigrid.cpp build_levels): the first level handled by the bottom of the cycle.  Rounds
    1-3 and SC_FLAG_BOTTOM_F32: the LDS-fit rule (bottom_start).  Default since round 4: the first level >= 2 with at most 127
    unknowns per side is the deepest one smoothed (k_mg_tail holds it in registers) and the level below it is the bottom; a ROI
    whose level 1 already fits the matrix-core solve (<= 96 per side) keeps that."""
    b = bottom_start(levels)
    if matrix_cores:
        nl = len(levels)
        a = next((l for l in range(2, nl - 1) if levels[l][0].n <= 127 and levels[l][1].n <= 127), 0)
        level1_direct = nl > 1 and levels[1][0].n <= 96 and levels[1][1].n <= 96
        if a and not (a == 2 and level1_direct):
            b = a + 1
        elif not level1_direct:
            b = next((l for l in range(1, nl) if levels[l][0].n <= 96 and levels[l][1].n <= 96), b)
    return b


def direct_level(levels, matrix_cores=True):
    """Index of the level solved exactly (fast diagonalisation), or None.  matrix_cores (the library's default since round 4):
    the bottom's first level is solved on the matrix cores whenever both sides have at most 96 unknowns, without an LDS budget
    to meet (sc_multigrid.cpp build_fd, k_mg_bottom_mm); False = SC_FLAG_BOTTOM_F32, the LDS-resident float32 form."""
    planes = 0
    b = bottom_level(levels, matrix_cores)
    for l in range(b, len(levels)):
        dx, dy = levels[l]
        planes += _bottom_floats(dx, dy)
        nxp, nyp = (dx.n + 3) // 4 * 4, (dy.n + 3) // 4 * 4
        if dx.n > 128 or dy.n > 128:
            continue
        if matrix_cores and l == b and dx.n <= 96 and dy.n <= 96:
            return l
        if 4 * (planes + 2 * nxp * nxp + 2 * nyp * nyp + 3 * nxp * nyp) > MG_BOTTOM_LDS_BYTES:
            continue
        return l
    return None


def solve_exact(F, dx: Dim, dy: Dim):
    """Exact solution of one level's system (zero ring) by a sparse direct solve in float64."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    def t1d(d: Dim):
        n = d.n
        main = np.full(n, -2.0); main[-1] = -float(d.d_last)
        sub = np.ones(n - 1); sup = np.ones(n - 1)
        if n >= 2:
            sub[-1] = float(d.cw_last)
        return sp.diags([sub, main, sup], [-1, 0, 1], format="csr") if n >= 2 else sp.csr_matrix(main.reshape(1, 1))
    Tx, Ty = t1d(dx), t1d(dy)
    A = sp.kron(sp.identity(dy.n), Tx) + sp.kron(Ty, sp.identity(dx.n))     # unknowns ordered y-major, x fastest
    u = spl.spsolve(A.tocsc(), F[1:-1, 1:-1].astype(np.float64).ravel())
    U = np.zeros_like(F)
    U[1:-1, 1:-1] = u.reshape(dy.n, dx.n).astype(F32)
    return U


def composes_level1(levels) -> bool:
    """The fused GPU path runs level 1 with 4 pre-smoothing sweeps and no post-smoothing when it is a launched level with a
    level 2 below it: the level-0 launch then interpolates from "level-1 correction + interpolated level-2 correction"
    directly and level 1 needs no prolongation launch (sc_multigrid.cpp: mg_composes_level1)."""
    return len(levels) >= 3 and bottom_level(levels) >= 2


def no_post_levels(levels):
    """Levels of the fused GPU schedule that run pre-smoothing only (sc_multigrid.cpp): level 1, where composes_level1()."""
    return (1,) if composes_level1(levels) else ()


def _f16(a):
    return a.astype(np.float16).astype(F32)


def _q16(a):
    """16-bit fixed point of the level-0 field between two launches (sc_cycle0.hip, TAG bits 8, 9):
    code = trunc(64 u + 16384.5) clamped to [0, 65535]; u = code / 64 - 256."""
    t = (a.astype(F32) * F32(64.0) + F32(16384.5)).astype(F32)
    code = np.clip(np.trunc(np.clip(t, 0.0, 65535.0)), 0, 65535).astype(F32)
    return (code * F32(0.015625) - F32(256.0)).astype(F32)


def vcycle(levels, l, U, F, pre=2, post=2, direct=None, no_post=(), level1_half=False, field_q16=False):
    """field_q16: the fast path stores level 0's field between its first launches (after the pre-smoothing; the residual uses the
    registers) as 16-bit fixed point (solve() says in which cycles).  level1_half: the library's fast path stores level 1's right-hand side (written by level 0's restriction) and level 1's
    smoothed correction (read by level 0's prolongation) as float16 (sc_cycle0.hip, TAG bit 7); level 1's own residual and
    restriction use the unrounded registers."""
    dx, dy = levels[l]
    if direct is not None and l == direct:
        return solve_exact(F, dx, dy)
    if l == len(levels) - 1:
        rho = 0.5 * (np.cos(np.pi / (dx.n + 1.0)) + np.cos(np.pi / (dy.n + 1.0)))
        om = 2.0 / (1.0 + np.sqrt(max(0.0, 1.0 - rho * rho)))
        return rb_gen(U, F, dx, dy, max(8, min(64, 2 * max(dx.n, dy.n))), float(F32(om)))
    # a level without post-smoothing does all its sweeps before the restriction
    U = rb_gen(U, F, dx, dy, pre + post if (l > 0 and l in no_post) else pre)
    Fc = restrict(residual_field(U, F, dx, dy), dx, dy)
    if level1_half and l == 0:
        Fc = _f16(Fc)
    E = vcycle(levels, l + 1, np.zeros_like(Fc), Fc, pre, post, direct, no_post, level1_half)
    U = _q16(U) if (field_q16 and l == 0) else U.copy()
    if level1_half and l == 1 and l in no_post:
        U = _f16(U)
    U += prolong(E, dx, dy)
    if l > 0 and l in no_post:
        return U
    return rb_gen(U, F, dx, dy, post)


def solve(U0, F, cycles=6, direct="auto", fused=True, level1_half=False, field_q16=False):
    """Multigrid solve of one plane: U0 carries the Dirichlet ring (level 0 is regular).
    direct: "auto" = the level the library solves directly, None = V-cycle down to the coarsest level.
    fused: the library's default (fused) schedule, in which level 1 has no post-smoothing where composes_level1();
    False = the textbook V(2,2) of the unfused path (sweeps_per_launch = 1)."""
    H, W = U0.shape
    levels = build_levels(W, H)
    d = direct_level(levels) if direct == "auto" else direct
    npl = no_post_levels(levels) if fused else ()
    U = U0.astype(F32).copy()
    for i in range(cycles):
        # the library rounds the first stores of a solve only: the launch before the judged cycle (the third, or the last) writes float
        U = vcycle(levels, 0, U, F, direct=d, no_post=npl, level1_half=level1_half and 1 in npl,
                   field_q16=field_q16 and level1_half and 1 in npl and i < min(cycles - 1, 2))
    return U
