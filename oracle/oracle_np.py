"""CPU oracle (numpy / scipy) for the seamless-clone hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seamlesscloneoptimization_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker.

What it restates (reference = wujinzhong/seamlessCloneOptimization, itself a restatement
of OpenCV 3.4.5 ``cv::seamlessClone(..., NORMAL_CLONE)``; OpenCV is an un-vendored
third-party dependency of the reference and is absent here):

  mask / ROI geometry   seamlessClone-CUDA/seamlessClone_imp.cpp:967-976 (border zero),
                        :927-963 + :1006-1016 (bounding box), :892-925 + :1060-1062
                        (3x 3x3 erode), :1066 (leftTop)
  gradients + blend     seamlessClone_imp.cpp:1920-1964
  divergence + fold     seamlessClone_imp.cpp:1966-2018
  direct DST-I solve    seamlessClone_imp.cpp:1814-1896 (OpenCV code quoted in comments
                        at :1342-1351, :1646-1655, :1825-1832), tables :581-599
  clamp/truncate/splice seamlessClone_imp.cpp:2078-2103 and :470-483

PARITY PIN STATUS: the reference ships no lossless golden output (its two result BMPs and
dst.yml are listed in .MISSING_LARGE_BLOBS).  This oracle is pinned by (a) the reference's
own input fixtures src.yml / src_mask.yml / sky.jpg, (b) its one committed output,
``blendedMat_0.jpg`` (lossy JPEG q95 4:2:0) -- see tests/test_oracle_golden.py -- and
(c) the exactness argument of SURVEY Appendix A.6 for the integer-valued stages.
Bit-level parity with OpenCV itself is therefore "parity unpinned"; what is pinned is
agreement with the reference's committed JPEG to within JPEG quantisation noise.

Two solver forms live here:
  * ``solve_dst``  -- float64 DST-I direct solve (what OpenCV / the reference compute);
  * ``jacobi`` / ``rbgs`` / ``residual`` -- float32 stencil sweeps with EXACTLY the
    operation order of the HIP kernels, so the GPU sweeps are checked bit-for-bit.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is only needed for the direct solve
    from scipy import fft as _sfft
except Exception:  # pragma: no cover
    _sfft = None

F32 = np.float32


# --------------------------------------------------------------------------------------
# A.1  mask stage / ROI geometry
# --------------------------------------------------------------------------------------
def zero_mask_border(mask: np.ndarray) -> np.ndarray:
    """seamlessClone_imp.cpp:967-976, launched with left=right=top=bottom=1, value 0 (:989)."""
    m = np.array(mask, dtype=np.uint8, copy=True)
    if m.ndim == 3:
        m = m[:, :, 0]
    m[0, :] = 0
    m[-1, :] = 0
    m[:, 0] = 0
    m[:, -1] = 0
    return m


def bounding_box(mask0: np.ndarray):
    """Bounding box {x0,x1,y0,y1} of mask!=0 (seamlessClone_imp.cpp:943-951).

    The host seeds {W-1, 0, H-1, 0} (:1006) so an empty mask yields x1<=x0 -> caller rejects
    (the reference asserts, :1013)."""
    h, w = mask0.shape
    ys, xs = np.nonzero(mask0)
    if xs.size == 0:
        return (w - 1, 0, h - 1, 0)
    return (int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()))


def erode3x3(m: np.ndarray) -> np.ndarray:
    """One pass of seamlessClone_imp.cpp:892-925: 255 iff the 9-sum == 255*9, frame forced 0."""
    h, w = m.shape
    out = np.zeros_like(m)
    if h < 3 or w < 3:
        return out
    s = np.zeros((h - 2, w - 2), dtype=np.int32)
    mi = m.astype(np.int32)
    for dy in range(3):
        for dx in range(3):
            s += mi[dy:dy + h - 2, dx:dx + w - 2]
    out[1:-1, 1:-1] = np.where(s == 255 * 9, 255, 0).astype(np.uint8)
    return out


def erode_min7(m: np.ndarray) -> np.ndarray:
    """OpenCV's erode(mask, 3x3 ones, iterations = 3) on the ROI VIEW of the zero-bordered mask, as cv::seamlessClone (OpenCV
    3.4.5, modules/photo/src/seamless_cloning_impl.cpp, Cloning::computeDerivatives) calls it: three 3x3 minimum filters = one
    7x7 minimum filter (cv::erode itself folds the iterations of a full rectangle into one kernel), reading through the
    view into the parent matrix, where everything outside the bounding box is zero (that is what makes it the bounding box)
    and the 1-pixel border was zeroed before; pixels outside the parent are ignored (morphologyDefaultBorderValue) but no
    window reaches them without also covering a zero.  Net: out = min over the 7x7 window with ZERO outside the ROI.
    For 0/255 masks this equals the reference's three thresholding passes (seamlessClone_imp.cpp:892-925); for grey masks the
    reference returns 0 wherever any of the 49 values is below 255, OpenCV returns their minimum.
    PARITY UNPINNED: OpenCV's source is not in /root/reference and no fixture of the reference holds a grey mask; this is a
    restatement of the published OpenCV 3.4.5 algorithm, used only behind SC_FLAG_OPENCV_GREY_MASK."""
    h, w = m.shape
    p = np.zeros((h + 6, w + 6), np.uint8)
    p[3:-3, 3:-3] = m
    out = np.full((h, w), 255, np.uint8)
    for dy in range(7):
        for dx in range(7):
            out = np.minimum(out, p[dy:dy + h, dx:dx + w])
    return out


def mask_stage(mask: np.ndarray, cx: int, cy: int, opencv_grey: bool = False):
    """Returns dict(x0,y0,W,H,ltx,lty,M) -- SURVEY Appendix A.1 steps 1-4.  opencv_grey: OpenCV's semantics for masks that
    are not 0/255 (erode_min7; build_rhs then blends with fractional weights) instead of the reference's thresholding."""
    m0 = zero_mask_border(mask)
    x0, x1, y0, y1 = bounding_box(m0)
    W = x1 - x0 + 1
    H = y1 - y0 + 1
    if not (x1 - x0 > 0 and y1 - y0 > 0):
        raise ValueError("empty mask (reference asserts at seamlessClone_imp.cpp:1013)")
    M = m0[y0:y0 + H, x0:x0 + W].copy()
    if opencv_grey:
        M = erode_min7(M)
    else:
        for _ in range(3):  # :1060-1062
            M = erode3x3(M)
    ltx = cx - (W >> 1)  # :1066
    lty = cy - (H >> 1)
    return dict(x0=x0, y0=y0, W=W, H=H, ltx=ltx, lty=lty, M=M, opencv_grey=bool(opencv_grey))


# --------------------------------------------------------------------------------------
# A.2 / A.3  gradients, blend, divergence, Dirichlet fold
# --------------------------------------------------------------------------------------
def _fwd_grad_reflect(I: np.ndarray):
    """Forward differences with reflect-101 at the last col/row (:1937,:1940,:1944,:1947)."""
    gx = np.empty_like(I)
    gy = np.empty_like(I)
    gx[:, :-1] = I[:, 1:] - I[:, :-1]
    gx[:, -1] = I[:, -2] - I[:, -1]
    gy[:-1, :] = I[1:, :] - I[:-1, :]
    gy[-1, :] = I[-2, :] - I[-1, :]
    return gx, gy


def build_rhs(dst: np.ndarray, patch: np.ndarray, geo: dict, dtype=F32):
    """Returns (B, lap, g).

    B   : H x W x 3  dst ROI as float (Dirichlet data lives on its ring)
    lap : H x W x 3  un-folded divergence of the blended gradient field, interior only
          (ring entries are 0) -- this is the stencil RHS of Appendix A.5
    g   : h x w x 3  compact folded RHS exactly as seamlessClone_imp.cpp:1992-2008
    Channel index = BGR index of the interleaved inputs (channels are independent)."""
    W, H, x0, y0, ltx, lty = (geo[k] for k in ("W", "H", "x0", "y0", "ltx", "lty"))
    if ltx < 0 or lty < 0 or ltx + W > dst.shape[1] or lty + H > dst.shape[0]:
        raise ValueError("ROI falls outside the destination image")
    B = dst[lty:lty + H, ltx:ltx + W, :].astype(dtype)
    P = patch[y0:y0 + H, x0:x0 + W, :].astype(dtype)
    m = (geo["M"].astype(dtype) * dtype(1.0 / 255.0))[:, :, None]  # :1950
    one = dtype(1.0)
    lap = np.zeros((H, W, 3), dtype=dtype)
    gxb, gyb = _fwd_grad_reflect(B)
    gxp, gyp = _fwd_grad_reflect(P)
    if geo.get("opencv_grey"):
        # OpenCV 3.4.5 Cloning::normalClone / evaluate: patchGradient * (M / 255) and destinationGradient * ((255 - M) / 255), each
        # weight a convertTo(CV_32F, 1.0 / 255.0) of the eroded mask / of its bitwise_not, then laplacianX = dest + patch
        mi = ((255 - geo["M"].astype(np.int32)).astype(dtype) * dtype(1.0 / 255.0))[:, :, None]
        GX = gxb * mi + gxp * m
        GY = gyb * mi + gyp * m
    else:
        GX = (one - m) * gxb + m * gxp  # :1952
        GY = (one - m) * gyb + m * gyp  # :1953
    lap[1:-1, 1:-1] = (GX[1:-1, 1:-1] - GX[1:-1, :-2]) + (GY[1:-1, 1:-1] - GY[:-2, 1:-1])  # :1987-1990
    g = lap[1:-1, 1:-1].copy()
    g[:, 0] -= B[1:-1, 0]      # x==1        :1992-1995
    g[0, :] -= B[0, 1:-1]      # y==1        :1996-1999
    g[:, -1] -= B[1:-1, -1]    # x==W-2      :2000-2003
    g[-1, :] -= B[-1, 1:-1]    # y==H-2      :2004-2007
    return B, lap, g


# --------------------------------------------------------------------------------------
# A.4  direct solve (float64 DST-I)
# --------------------------------------------------------------------------------------
PI_F = float(np.float32(3.14159265358979323846))     # the reference's `#define PI ...f` (seamlessClone_imp.h:17)


def solve_dst(g: np.ndarray, float_tables: bool = False) -> np.ndarray:
    """u = DST1^-1( DST1(g) / (2cos(pi(i+1)/(w+1)) + 2cos(pi(j+1)/(h+1)) - 4) ), per channel.

    seamlessClone_imp.cpp:1825-1832 (divide), :596-599 (filter_X/Y tables).
    float_tables=False: the exact linear system (denominator in float64).
    float_tables=True : the reference's own denominator -- tables computed in double from the float literal PI
    but STORED as float (:596-599) and summed in float (:1651-1653); the transforms stay float64.  The two answers
    differ by up to 3 grey levels at 2048^2 and 7 at 4096^2 (low modes only)."""
    if _sfft is None:  # pragma: no cover
        raise RuntimeError("scipy required for solve_dst")
    g64 = g.astype(np.float64)
    h, w = g64.shape[:2]
    if float_tables:
        fx = (2.0 * np.cos(PI_F / (w + 1.0) * (np.arange(w) + 1.0))).astype(F32)
        fy = (2.0 * np.cos(PI_F / (h + 1.0) * (np.arange(h) + 1.0))).astype(F32)
        den = ((fx[None, :] + fy[:, None]).astype(F32) - F32(4.0)).astype(np.float64)
    else:
        fx = 2.0 * np.cos(np.pi * (np.arange(w) + 1.0) / (w + 1.0))
        fy = 2.0 * np.cos(np.pi * (np.arange(h) + 1.0) / (h + 1.0))
        den = (fx[None, :] + fy[:, None] - 4.0)
    out = np.empty_like(g64)
    for c in range(g64.shape[2]):
        t = _sfft.dstn(g64[:, :, c], type=1)
        out[:, :, c] = _sfft.idstn(t / den, type=1)
    return out


# --------------------------------------------------------------------------------------
# A.5  stencil sweeps, float32, GPU operation order
# --------------------------------------------------------------------------------------
def jacobi(U: np.ndarray, lap: np.ndarray, sweeps: int) -> np.ndarray:
    """U' = 0.25f * (((l + r) + (u + d)) - f) on the interior; ring held fixed.  U: H x W (x C)."""
    U = U.astype(F32, copy=True)
    lap = lap.astype(F32, copy=False)
    q = F32(0.25)
    for _ in range(sweeps):
        s = (U[1:-1, :-2] + U[1:-1, 2:]) + (U[:-2, 1:-1] + U[2:, 1:-1])
        V = U.copy()
        V[1:-1, 1:-1] = q * (s - lap[1:-1, 1:-1])
        U = V
    return U


def _color_mask(H, W, color):
    yy, xx = np.mgrid[0:H, 0:W]
    m = ((xx + yy) & 1) == color
    m[0, :] = m[-1, :] = False
    m[:, 0] = m[:, -1] = False
    return m


def rbgs(U: np.ndarray, lap: np.ndarray, sweeps: int, omega: float = 1.0) -> np.ndarray:
    """Red-black Gauss-Seidel / SOR, colour = (x+y)&1 in ROI coordinates, colour 0 first.

    gs = 0.25f*(((l+r)+(u+d)) - f);  U = U + omega*(gs - U)   (no FMA; omega==1 -> U = gs)."""
    U = U.astype(F32, copy=True)
    lap = lap.astype(F32, copy=False)
    H, W = U.shape[:2]
    q = F32(0.25)
    om = F32(omega)
    masks = [_color_mask(H, W, 0), _color_mask(H, W, 1)]
    for _ in range(sweeps):
        for cm in masks:
            s = np.zeros_like(U)
            s[1:-1, 1:-1] = (U[1:-1, :-2] + U[1:-1, 2:]) + (U[:-2, 1:-1] + U[2:, 1:-1])
            gs = q * (s - lap)
            if omega == 1.0:
                new = gs
            else:
                new = U + om * (gs - U)
            U[cm] = new[cm]
    return U


def residual(U: np.ndarray, lap: np.ndarray):
    """(sum r^2, sum lap^2) over interior+channels, r = lap - ((l+r)+(u+d) - 4U), float64 sums
    of float32 point values."""
    U = U.astype(F32, copy=False)
    lap = lap.astype(F32, copy=False)
    s = (U[1:-1, :-2] + U[1:-1, 2:]) + (U[:-2, 1:-1] + U[2:, 1:-1])
    r = lap[1:-1, 1:-1] - (s - F32(4.0) * U[1:-1, 1:-1])
    return float(np.sum(r.astype(np.float64) ** 2)), float(np.sum(lap[1:-1, 1:-1].astype(np.float64) ** 2))


# --------------------------------------------------------------------------------------
# A.7  output
# --------------------------------------------------------------------------------------
def clamp_truncate(u: np.ndarray) -> np.ndarray:
    """clamp to [0,255] then C-style truncation toward zero (seamlessClone_imp.cpp:2091-2094)."""
    return np.clip(u, 0.0, 255.0).astype(np.uint8)


def splice(dst: np.ndarray, u8: np.ndarray, geo: dict) -> np.ndarray:
    """Write the h x w interior at (lty+1, ltx+1); ring and outside keep dst (:470-483)."""
    out = dst.copy()
    H, W, ltx, lty = geo["H"], geo["W"], geo["ltx"], geo["lty"]
    out[lty + 1:lty + H - 1, ltx + 1:ltx + W - 1, :] = u8
    return out


def seamless_clone(dst, patch, mask, cx, cy, return_all=False, float_tables=False, opencv_grey=False):
    """Full NORMAL_CLONE path with the float64 direct solve.  dst HxWx3 u8 BGR (any channel
    order works, channels are independent), patch hxwx3 u8, mask hxw u8.  float_tables: see solve_dst
    (True = the reference's arithmetic, which the HIP library reproduces by default).  opencv_grey: see mask_stage."""
    geo = mask_stage(mask, cx, cy, opencv_grey)
    B, lap, g = build_rhs(dst, patch, geo, dtype=np.float64)
    u = solve_dst(g, float_tables)
    out = splice(dst, clamp_truncate(u), geo)
    if return_all:
        return out, dict(geo=geo, B=B, lap=lap, g=g, u=u)
    return out


def full_field(B: np.ndarray, u_int: np.ndarray) -> np.ndarray:
    """Ring from B, interior from u_int -> H x W x C field (Appendix A.5 layout)."""
    U = B.copy()
    U[1:-1, 1:-1] = u_int
    return U


# --------------------------------------------------------------------------------------
# synthetic inputs -- the same generator is used by tests and bench
# --------------------------------------------------------------------------------------
def synth_inputs(W: int, H: int, seed_dst: int = 1001, seed_patch: int = 2002, margin: int = 256,
                 ellipse: bool = False):
    """dst (H+margin)x(W+margin)x3, patch/mask (H+2)x(W+2): a W x H ROI after border zeroing."""
    Hd, Wd = H + margin, W + margin
    rng = np.random.default_rng(seed_dst)
    yy, xx = np.mgrid[0:Hd, 0:Wd]
    base = 128.0 + 60.0 * np.sin(2 * np.pi * xx / Wd) * np.cos(2 * np.pi * yy / Hd)
    dst = np.clip(base[:, :, None] + rng.normal(0.0, 12.0, (Hd, Wd, 3)), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(seed_patch)
    Hp, Wp = H + 2, W + 2
    yy, xx = np.mgrid[0:Hp, 0:Wp]
    base = 110.0 + 50.0 * np.cos(3 * np.pi * xx / max(W, 1))
    patch = np.clip(base[:, :, None] + rng.normal(0.0, 20.0, (Hp, Wp, 3)), 0, 255).astype(np.uint8)
    mask = np.full((Hp, Wp), 255, np.uint8)
    if ellipse:
        cy0, cx0 = (Hp - 1) / 2.0, (Wp - 1) / 2.0
        mask = np.where(((yy - cy0) / (Hp / 2.0 - 1)) ** 2 + ((xx - cx0) / (Wp / 2.0 - 1)) ** 2 <= 1.0,
                        255, 0).astype(np.uint8)
    return dst, patch, mask, Wd // 2, Hd // 2
