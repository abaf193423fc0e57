"""GPU parity tests added in round 5 (through the C ABI, against the CPU oracle): the scan's completion fence without stage
marks, the host-image call's ROI-only return under concurrent writers, groups of clones with different ROI sizes."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracles():
    from oracle import oracle_np, oracle_c
    oracle_c.build()
    return oracle_np, oracle_c


@pytest.fixture()
def inst():
    """A fresh instance with the library's DEFAULT options (the session fixture `hip` is pinned to multigrid)."""
    from seamlesscloneoptimization_amd import capi
    i = capi.Instance(0)
    yield i
    i.destroy()


def _dmax(a, b):
    return int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max())


@pytest.mark.parametrize("flags_name", ["NO_STAGE_MARKS", "NO_STAGE_MARKS|ROWS_RETURN", "default"])
def test_changed_mask_without_stage_marks_is_noticed(inst, oracles, flags_name):
    """with SC_FLAG_NO_STAGE_MARKS the host-image call used stage mark 5 -- which that flag never records --
    as the completion fence of the bounding-box scan that rides in the pre-process launch; the wait returned at once, the
    comparison read the PREVIOUS call's rectangle, which equals the remembered guess, and a call whose mask had changed at the
    same size returned SC_OK with nothing cloned (the direct solve has no other host wait before that point).  The fence is an
    event of its own now.  AUTO on a 300 x 200 mask = the direct FFT solve; every call must match the oracle on ITS mask and
    report ITS geometry."""
    from seamlesscloneoptimization_amd import capi
    o, _ = oracles
    flags = {"NO_STAGE_MARKS": capi.SC_FLAG_NO_STAGE_MARKS, "NO_STAGE_MARKS|ROWS_RETURN": capi.SC_FLAG_NO_STAGE_MARKS | capi.SC_FLAG_ROWS_RETURN,
             "default": 0}[flags_name]
    inst.set_solver(flags=flags)
    rng = np.random.default_rng(11)
    H, W = 200, 300
    dst = rng.integers(0, 256, (H + 60, W + 60, 3), dtype=np.uint8)
    patch = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    full = np.full((H, W), 255, np.uint8)
    blob = np.zeros((H, W), np.uint8); blob[30:170, 40:220] = 255
    other = np.zeros((H, W), np.uint8); other[60:190, 100:290] = 255
    cx, cy = (W + 60) // 2, (H + 60) // 2
    for k, m in enumerate([full, full, blob, blob, other, full, other, other, blob]):
        want = o.seamless_clone(dst, patch, m, cx, cy, float_tables=True)
        body = dst.copy()
        assert inst.run(patch, body, m, cx, cy) == 0
        i = inst.info()
        assert i.method == capi.SC_METHOD_FFT
        ys, xs = np.nonzero(m[1:-1, 1:-1])
        assert (i.x0, i.y0, i.W, i.H) == (xs.min() + 1, ys.min() + 1, xs.max() - xs.min() + 1, ys.max() - ys.min() + 1), k
        assert _dmax(want, body) <= 1, k
        assert not np.array_equal(body, dst), k


def test_two_host_calls_into_disjoint_rois_of_one_destination(oracles):
    """the default return of the host-image call writes ROI bytes only (as the reference's splice does,
    seamlessClone_imp.cpp:470-483), so two calls that clone into disjoint column ranges of ONE destination at the same time --
    two threads with an instance each, and two host jobs of one pool -- both keep their result.  (Late round 4's default wrote
    whole rows back: the later call erased the earlier one's ROI.)"""
    from seamlesscloneoptimization_amd import capi
    o, oc = oracles
    rng = np.random.default_rng(3)
    Hd, Wd = 700, 1500                        # the two ROIs cover most of a row together, so each would qualify for the row return alone
    dst = rng.integers(0, 256, (Hd, Wd, 3), dtype=np.uint8)
    pa = rng.integers(0, 256, (640, 700, 3), dtype=np.uint8)
    pb = rng.integers(0, 256, (640, 700, 3), dtype=np.uint8)
    mask = np.full((640, 700), 255, np.uint8)
    ca, cb = (370, 350), (1120, 350)
    want = oc.seamless_clone(dst, pa, mask, ca[0], ca[1], min(16, oc.max_threads()))
    want = oc.seamless_clone(want, pb, mask, cb[0], cb[1], min(16, oc.max_threads()))      # disjoint ROIs: the order does not matter
    # (a) two threads, one instance each
    for rep in range(3):
        body = dst.copy()
        insts = [capi.Instance(0), capi.Instance(0)]
        try:
            bar = threading.Barrier(2)
            rcs = [None, None]

            def work(k, patch, c):
                bar.wait()
                rcs[k] = insts[k].run(patch, body, mask, c[0], c[1])
            th = [threading.Thread(target=work, args=(0, pa, ca)), threading.Thread(target=work, args=(1, pb, cb))]
            for t in th: t.start()
            for t in th: t.join()
        finally:
            for i in insts: i.destroy()
        assert rcs == [0, 0]
        assert _dmax(want, body) <= 1, rep
    # (b) two host jobs of one pool sharing the destination
    pool = capi.Pool(0, streams=2)
    try:
        body = dst.copy()
        pool.run_host([(pa, body, mask, ca[0], ca[1]), (pb, body, mask, cb[0], cb[1])])
        assert _dmax(want, body) <= 1
    finally:
        pool.close()
