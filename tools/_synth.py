"""Synthetic inputs of SURVEY 8d for the measurement scripts (numpy only; the product tools do not import oracle/)."""
import numpy as np


def synth_inputs(W, H, seed_dst=1001, seed_patch=2002, margin=256):
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
    return dst, patch, mask, Wd // 2, Hd // 2
