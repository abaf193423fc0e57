import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import oracle_np as o, mg_np as mg
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = W
kind = sys.argv[2] if len(sys.argv) > 2 else "std"
dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=32)
if kind == "noise":
    rng = np.random.default_rng(1); patch = rng.integers(0, 256, patch.shape, dtype=np.uint8); dst = rng.integers(0, 256, dst.shape, dtype=np.uint8)
geo = o.mask_stage(mask, cx, cy)
B, lap, g = o.build_rhs(dst, patch, geo, dtype=np.float64)
c = 1
uex = o.full_field(B, o.solve_dst(g))[:, :, c]
levels = mg.build_levels(W, H)
d = mg.direct_level(levels)
npl = mg.no_post_levels(levels)
U0 = B[:, :, c].astype(np.float32); F = lap[:, :, c].astype(np.float32)
# plain V-cycles from the destination
U = U0.copy()
for k in range(4):
    U = mg.vcycle(levels, 0, U, F, direct=d, no_post=npl)
    print("plain cycle", k + 1, "max err %.4f" % np.abs(U - uex).max(), flush=True)
# FMG on the error equation: A e = r0 = f - A u0, zero Dirichlet
dx, dy = levels[0]
r0 = mg.residual_field(U0, F, dx, dy)
Fs = [r0]
for l in range(len(levels) - 1):
    Fs.append(mg.restrict(Fs[-1], *levels[l]))
L = d if d is not None else len(levels) - 1
e = mg.solve_exact(Fs[L], *levels[L]) if d is not None else None
for l in range(L - 1, -1, -1):
    e = mg.prolong(e, *levels[l])
    if l > 0:
        e = mg.vcycle(levels, l, e, Fs[l], direct=d, no_post=())
eex = uex - U0
print("FMG interpolated to level 0: max err %.4f" % np.abs(e - eex).max())
for k in range(3):
    e = mg.vcycle(levels, 0, e, Fs[0], direct=d, no_post=npl)
    print("FMG + cycle", k + 1, "max err %.4f" % np.abs(e - eex).max(), flush=True)
