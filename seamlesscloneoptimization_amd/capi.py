"""ctypes binding of libseamlessclone_hip.so -- the C ABI declared in include/seamlessclone_hip.h.

There is no CPU fallback: if the HIP library is missing or does not load this module raises,
and every compute entry point needs a visible MI355X.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libseamlessclone_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "seamlessclone_hip.h")

SC_OK = 0
SC_ERR_BAD_ARG = -1
SC_ERR_BAD_SIZE = -2
SC_ERR_EMPTY_MASK = -3
SC_ERR_ROI_OOB = -4
SC_ERR_HIP = -5
SC_ERR_NOT_CONVERGED = -6

SC_METHOD_JACOBI = 0
SC_METHOD_RBGS = 1
SC_METHOD_SOR = 2
SC_METHOD_MULTIGRID = 3
SC_METHOD_DST = 4
SC_METHOD_AUTO = 5      # default: the FFT-form direct solve (double transforms) up to SC_AUTO_DIRECT_MAX unknowns per side, MULTIGRID above
SC_METHOD_FFT = 6       # the reference's default back-end: FFT-based direct solve, float32, O(n^2 log n)
SC_AUTO_DIRECT_MAX = 720
SC_AUTO_DIRECT_AREA = 450000
SC_AUTO_NARROW_MAX = 140
SC_AUTO_THIN_MAX = 4
SC_AUTO_THIN_LONG_MAX = 4096

SC_FLAG_NO_SPECULATE = 1 << 0
SC_FLAG_FLOAT_RHS = 1 << 1
SC_FLAG_FLOAT_U0 = 1 << 2
SC_FLAG_NO_COMPOSE_L1 = 1 << 3
SC_FLAG_VCYCLE_BOTTOM = 1 << 4
SC_FLAG_EXACT_TABLES = 1 << 5
SC_FLAG_LEGACY_PATHS = 1 << 6      # run the superseded launch forms named in SolverOpts.legacy_paths (SC_LEGACY_* bits)
SC_LEGACY_SEPARATE_RESTRICT, SC_LEGACY_BOTTOM_F32, SC_LEGACY_SEPARATE_TAIL = 1, 2, 4
SC_FLAG_KEEP_FIELD = 1 << 7
SC_FLAG_FFT_FP64 = 1 << 8
SC_FLAG_OPENCV_GREY_MASK = 1 << 9
SC_FLAG_FLOAT_L1 = 1 << 10
SC_FLAG_FLOAT_FIELD = 1 << 11
SC_FLAG_NO_STAGE_MARKS = 1 << 12
SC_FLAG_ROWS_RETURN = 1 << 13     # host-image call, opt-in: whole destination rows come back as one linear copy (default: ROI bytes only)
SC_FLAG_POISON_ARENA = 1 << 14    # testing: blocks handed out without zeroing are filled with 0xFF (NaN) first

def auto_takes_direct(w: int, h: int) -> bool:
    """SC_METHOD_AUTO's choice for ONE clone with w x h unknowns (sc_solver.cpp effective_method): True = the direct solve (SC_METHOD_FFT,
    double transforms), False = multigrid.  (A group of clones always takes the cycles; tol > 0 too.)"""
    if w <= SC_AUTO_DIRECT_MAX and h <= SC_AUTO_DIRECT_MAX:
        return True
    if max(w, h) <= SC_AUTO_THIN_LONG_MAX and (w * h <= SC_AUTO_DIRECT_AREA or min(w, h) <= SC_AUTO_NARROW_MAX):
        return True
    return False


ERR_NAMES = {
    SC_ERR_BAD_ARG: "SC_ERR_BAD_ARG", SC_ERR_BAD_SIZE: "SC_ERR_BAD_SIZE", SC_ERR_EMPTY_MASK: "SC_ERR_EMPTY_MASK",
    SC_ERR_ROI_OOB: "SC_ERR_ROI_OOB", SC_ERR_HIP: "SC_ERR_HIP", SC_ERR_NOT_CONVERGED: "SC_ERR_NOT_CONVERGED",
}


class SolverOpts(C.Structure):
    _fields_ = [("method", C.c_int), ("max_sweeps", C.c_int), ("tol", C.c_float), ("check_every", C.c_int),
                ("omega", C.c_float), ("sweeps_per_launch", C.c_int), ("reference_warmup", C.c_int),
                ("mg_pre", C.c_int), ("mg_post", C.c_int), ("update_tol", C.c_float), ("flags", C.c_int), ("jacobi_tile_rows", C.c_int), ("mg_level1_sweeps", C.c_int), ("mg_direct_max", C.c_int), ("legacy_paths", C.c_int)]


class RunInfo(C.Structure):
    _fields_ = [("x0", C.c_int), ("y0", C.c_int), ("W", C.c_int), ("H", C.c_int), ("ltx", C.c_int), ("lty", C.c_int),
                ("sweeps", C.c_int), ("converged", C.c_int), ("rel_residual", C.c_double),
                ("ms_h2d", C.c_float), ("ms_mask", C.c_float), ("ms_pre", C.c_float), ("ms_solve", C.c_float),
                ("ms_post", C.c_float), ("ms_d2h", C.c_float), ("ms_device_total", C.c_float),
                ("sweep_launches", C.c_int), ("last_update", C.c_float), ("device_bytes", C.c_size_t), ("method", C.c_int),
                ("device", C.c_int), ("ms_call", C.c_float), ("field_retry", C.c_int), ("new_size", C.c_int),
                ("group_members", C.c_int), ("group_ragged", C.c_int)]


class BatchJob(C.Structure):
    _fields_ = [("face", C.c_void_p), ("face_cols", C.c_int), ("face_rows", C.c_int), ("face_step", C.c_int),
                ("body", C.c_void_p), ("body_cols", C.c_int), ("body_rows", C.c_int), ("body_step", C.c_int),
                ("mask", C.c_void_p), ("mask_cols", C.c_int), ("mask_rows", C.c_int), ("mask_step", C.c_int),
                ("centerX", C.c_int), ("centerY", C.c_int), ("body_restore", C.c_void_p), ("rc", C.c_int)]


class SeamlessCloneError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")


u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
f64p = C.POINTER(C.c_double)
_IMG = [C.c_void_p, C.c_int, C.c_int, C.c_int]

_lib = None


def declared_symbols() -> list[str]:
    """Every function name the public header declares (used by the CPU-side ABI test)."""
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"SC_API[^;(]*?\b((?:my_seamlessclone_api_imp_|sc_hip_)\w+)\s*\(", txt)))


def build(verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (no GPU needed)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.my_seamlessclone_api_imp_create_instance.argtypes = [C.c_int]
    L.my_seamlessclone_api_imp_create_instance.restype = C.c_void_p
    L.my_seamlessclone_api_imp_run.argtypes = [C.c_void_p] + _IMG * 3 + [C.c_int, C.c_int, C.c_int, C.c_bool]
    L.my_seamlessclone_api_imp_run.restype = C.c_int
    L.my_seamlessclone_api_imp_destroy.argtypes = [C.c_void_p]
    L.my_seamlessclone_api_imp_destroy.restype = None
    L.my_seamlessclone_api_imp_sync.argtypes = [C.c_void_p]
    L.my_seamlessclone_api_imp_sync.restype = None
    L.sc_hip_default_opts.argtypes = [C.POINTER(SolverOpts)]
    L.sc_hip_default_opts.restype = None
    L.sc_hip_set_solver.argtypes = [C.c_void_p, C.POINTER(SolverOpts)]
    L.sc_hip_set_solver.restype = C.c_int
    L.sc_hip_get_solver.argtypes = [C.c_void_p, C.POINTER(SolverOpts)]
    L.sc_hip_get_solver.restype = C.c_int
    L.sc_hip_get_info.argtypes = [C.c_void_p, C.POINTER(RunInfo)]
    L.sc_hip_get_info.restype = C.c_int
    L.sc_hip_last_error.argtypes = [C.c_void_p]
    L.sc_hip_last_error.restype = C.c_char_p
    L.sc_hip_run_device.argtypes = [C.c_void_p] + _IMG * 3 + [C.c_int, C.c_int, C.c_bool]
    L.sc_hip_run_device.restype = C.c_int
    L.sc_hip_malloc.argtypes = [C.c_void_p, C.c_size_t]
    L.sc_hip_malloc.restype = C.c_void_p
    L.sc_hip_free.argtypes = [C.c_void_p, C.c_void_p]
    L.sc_hip_free.restype = None
    L.sc_hip_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.sc_hip_host_alloc.restype = C.c_void_p
    L.sc_hip_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.sc_hip_host_free.restype = None
    L.sc_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sc_hip_memcpy_h2d.restype = C.c_int
    L.sc_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sc_hip_memcpy_d2h.restype = C.c_int
    L.sc_hip_memcpy_d2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sc_hip_memcpy_d2d_async.restype = C.c_int
    L.sc_hip_device_count.argtypes = []
    L.sc_hip_device_count.restype = C.c_int
    L.sc_hip_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.sc_hip_device_pci_bus_id.restype = C.c_int
    L.sc_hip_mask_stage.argtypes = [C.c_void_p] + _IMG + [C.c_int, C.c_int, i32p, u8p, C.c_size_t]
    L.sc_hip_mask_stage.restype = C.c_int
    L.sc_hip_build_rhs.argtypes = [C.c_void_p] + _IMG * 3 + [C.c_int, C.c_int, i32p, f32p, f32p, C.c_size_t]
    L.sc_hip_build_rhs.restype = C.c_int
    L.sc_hip_field_load.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p]
    L.sc_hip_field_load.restype = C.c_int
    L.sc_hip_field_sweep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
    L.sc_hip_field_sweep.restype = C.c_int
    L.sc_hip_field_residual.argtypes = [C.c_void_p, f64p]
    L.sc_hip_field_residual.restype = C.c_int
    L.sc_hip_field_solve.argtypes = [C.c_void_p]
    L.sc_hip_field_solve.restype = C.c_int
    L.sc_hip_field_shape.argtypes = [C.c_void_p, i32p]
    L.sc_hip_field_shape.restype = C.c_int
    L.sc_hip_field_store.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.sc_hip_field_store.restype = C.c_int
    L.sc_hip_field_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.sc_hip_field_finish.restype = C.c_int
    L.sc_hip_field_lowmode.argtypes = [C.c_void_p]
    L.sc_hip_field_lowmode.restype = C.c_int
    L.sc_hip_field_time_sweeps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    L.sc_hip_field_time_sweeps.restype = C.c_int
    L.sc_hip_pool_create.argtypes = [C.c_int, C.c_int]
    L.sc_hip_pool_create.restype = C.c_void_p
    L.sc_hip_pool_destroy.argtypes = [C.c_void_p]
    L.sc_hip_pool_destroy.restype = None
    L.sc_hip_pool_size.argtypes = [C.c_void_p]
    L.sc_hip_pool_size.restype = C.c_int
    L.sc_hip_pool_instance.argtypes = [C.c_void_p, C.c_int]
    L.sc_hip_pool_instance.restype = C.c_void_p
    L.sc_hip_pool_set_solver.argtypes = [C.c_void_p, C.POINTER(SolverOpts)]
    L.sc_hip_pool_set_solver.restype = C.c_int
    L.sc_hip_pool_run.argtypes = [C.c_void_p, C.POINTER(BatchJob), C.c_int, C.c_int]
    L.sc_hip_pool_run.restype = C.c_int
    L.sc_hip_pool_set_group.argtypes = [C.c_void_p, C.c_int]
    L.sc_hip_pool_set_group.restype = C.c_int
    L.sc_hip_run_device_batch.argtypes = [C.c_void_p, C.POINTER(BatchJob), C.c_int]
    L.sc_hip_run_device_batch.restype = C.c_int
    L.sc_hip_time_cycle0.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.sc_hip_time_cycle0.restype = C.c_int
    L.sc_hip_time_coarse_chain.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.sc_hip_time_coarse_chain.restype = C.c_int
    L.sc_hip_time_tail_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.sc_hip_time_tail_phases.restype = C.c_int
    L.sc_hip_time_cycle0_form.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.sc_hip_time_cycle0_form.restype = C.c_int
    L.sc_hip_reference_tables_singular.argtypes = [C.c_int, C.c_int]
    L.sc_hip_reference_tables_singular.restype = C.c_int
    L.sc_hip_selftest_host.argtypes = []
    L.sc_hip_selftest_host.restype = C.c_int
    L.sc_hip_plan_groups.argtypes = [i32p, C.c_int, C.c_int, C.POINTER(SolverOpts), i32p, i32p]
    L.sc_hip_plan_groups.restype = C.c_int
    L.sc_hip_plan_size.argtypes = [C.c_int, C.c_int, C.POINTER(SolverOpts), i32p]
    L.sc_hip_plan_groups_pool.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.POINTER(SolverOpts), i32p, i32p]
    L.sc_hip_plan_groups_pool.restype = C.c_int
    L.sc_hip_plan_prepare.argtypes = [i32p, C.c_int, C.POINTER(SolverOpts)]
    L.sc_hip_plan_prepare.restype = C.c_int
    L.sc_hip_plan_cache_clear.argtypes = []
    L.sc_hip_plan_cache_clear.restype = None
    L.sc_hip_plan_size.restype = C.c_int
    _lib = L
    return L


def _img(a: np.ndarray):
    """numpy HxW[x3] uint8 (row-contiguous, arbitrary row stride) -> (ptr, cols, rows, step)."""
    if a.dtype != np.uint8:
        raise TypeError("images must be uint8")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    ch = 1 if a.ndim == 2 else a.shape[2]
    if a.strides[-1] != 1 or (a.ndim == 3 and a.strides[1] != ch):
        raise ValueError("image rows must be contiguous (cv::Mat layout)")
    return a.ctypes.data, a.shape[1], a.shape[0], a.strides[0]


class Instance:
    """Thin RAII wrapper over one library instance (one GPU, one stream)."""

    def __init__(self, gpu_id: int = 0):
        self.L = load()
        self.h = self.L.my_seamlessclone_api_imp_create_instance(int(gpu_id))
        if not self.h:
            raise SeamlessCloneError(SC_ERR_HIP, f"cannot create an instance on GPU {gpu_id} "
                                     "(no MI355X visible? there is no CPU fallback)")
        self.gpu_id = gpu_id

    # ---- lifetime
    def destroy(self):
        if getattr(self, "h", None):
            self.L.my_seamlessclone_api_imp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def sync(self):
        self.L.my_seamlessclone_api_imp_sync(self.h)

    def _check(self, rc, allow=()):
        if rc != SC_OK and rc not in allow:
            raise SeamlessCloneError(rc, (self.L.sc_hip_last_error(self.h) or b"").decode())
        return rc

    # ---- options / info
    def default_opts(self) -> SolverOpts:
        o = SolverOpts()
        self.L.sc_hip_default_opts(C.byref(o))
        return o

    def set_solver(self, **kw) -> SolverOpts:
        o = self.get_solver()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        self._check(self.L.sc_hip_set_solver(self.h, C.byref(o)))
        return o

    def get_solver(self) -> SolverOpts:
        o = SolverOpts()
        self._check(self.L.sc_hip_get_solver(self.h, C.byref(o)))
        return o

    def info(self) -> RunInfo:
        i = RunInfo()
        self._check(self.L.sc_hip_get_info(self.h, C.byref(i)))
        return i

    # ---- the clone
    def run(self, face, body, mask, cx, cy, sync=False, allow_not_converged=False):
        """In place on `body` (reference semantics).  Returns the C return code.  The call is complete when it returns
        either way; sync = the reference's bSync: True also prints its two timing lines on stdout (the reference's
        Python binding passes False, SeamlessClone.cpp:63)."""
        if not body.flags.writeable:
            raise ValueError("body must be writeable: the clone is in place")
        f, b, m = _img(face), _img(body), _img(mask)
        rc = self.L.my_seamlessclone_api_imp_run(self.h, *f, *b, *m, int(cx), int(cy), self.gpu_id, bool(sync))
        return self._check(rc, allow=(SC_ERR_NOT_CONVERGED,) if allow_not_converged else ())

    # ---- device-resident images
    def malloc(self, nbytes):
        p = self.L.sc_hip_malloc(self.h, nbytes)
        if not p:
            raise SeamlessCloneError(SC_ERR_HIP, f"hipMalloc({nbytes}) failed")
        return p

    def free(self, p):
        self.L.sc_hip_free(self.h, p)

    def pinned_array(self, shape, dtype=np.uint8):
        """numpy array over page-locked host memory (hipHostMalloc).  Returns (array, handle); release the memory with
        free_pinned(handle) after the last use of the array."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.sc_hip_host_alloc(self.h, n)
        if not p:
            raise SeamlessCloneError(SC_ERR_HIP, f"hipHostMalloc({n}) failed")
        buf = (C.c_uint8 * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape), p

    def free_pinned(self, handle):
        self.L.sc_hip_host_free(self.h, handle)

    def to_device(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        p = self.malloc(a.nbytes)
        self._check(self.L.sc_hip_memcpy_h2d(self.h, p, a.ctypes.data, a.nbytes))
        return p

    def from_device(self, p, shape, dtype=np.uint8):
        out = np.empty(shape, dtype)
        self._check(self.L.sc_hip_memcpy_d2h(self.h, out.ctypes.data, p, out.nbytes))
        return out

    def copy_d2d_async(self, dst, src, nbytes):
        self._check(self.L.sc_hip_memcpy_d2d_async(self.h, dst, src, nbytes))

    def run_device(self, d_face, fshape, d_body, bshape, d_mask, mshape, cx, cy, sync=True,
                   allow_not_converged=False):
        """shapes are (rows, cols); rows are dense (step = cols * channels)."""
        rc = self.L.sc_hip_run_device(self.h, d_face, fshape[1], fshape[0], 3 * fshape[1],
                                      d_body, bshape[1], bshape[0], 3 * bshape[1],
                                      d_mask, mshape[1], mshape[0], mshape[1], int(cx), int(cy), bool(sync))
        return self._check(rc, allow=(SC_ERR_NOT_CONVERGED,) if allow_not_converged else ())

    def run_device_batch(self, jobs, sync=True):
        """jobs: a BatchJob array (Pool.make_jobs) of device-resident clones.  Members whose ROIs have one size are solved
        as one field of 3n channels through one set of launches (sc_hip_run_device_batch); per-member codes in jobs[i].rc."""
        rc = self.L.sc_hip_run_device_batch(self.h, jobs, len(jobs))
        if sync:
            self.sync()
        return self._check(rc, allow=(SC_ERR_NOT_CONVERGED,))

    # ---- stage hooks
    def mask_stage(self, mask, cx, cy):
        m = _img(mask)
        geo = np.zeros(6, np.int32)
        M = np.zeros(m[1] * m[2], np.uint8)
        self._check(self.L.sc_hip_mask_stage(self.h, *m, int(cx), int(cy), geo.ctypes.data_as(i32p),
                                             M.ctypes.data_as(u8p), M.size))
        W, H = int(geo[2]), int(geo[3])
        return geo, M[:W * H].reshape(H, W).copy()

    def build_rhs(self, face, body, mask, cx, cy):
        f, b, m = _img(face), _img(body), _img(mask)
        geo = np.zeros(6, np.int32)
        cap = m[1] * m[2]
        B = np.zeros(3 * cap, np.float32)
        lap = np.zeros(3 * cap, np.float32)
        self._check(self.L.sc_hip_build_rhs(self.h, *f, *b, *m, int(cx), int(cy), geo.ctypes.data_as(i32p),
                                            B.ctypes.data_as(f32p), lap.ctypes.data_as(f32p), cap))
        W, H = int(geo[2]), int(geo[3])
        return geo, B[:3 * W * H].reshape(3, H, W).copy(), lap[:3 * W * H].reshape(3, H, W).copy()

    def field_load(self, U, lap):
        U = np.ascontiguousarray(U, np.float32)
        lap = np.ascontiguousarray(lap, np.float32)
        assert U.shape == lap.shape and U.ndim == 3
        Cc, H, W = U.shape
        self._check(self.L.sc_hip_field_load(self.h, W, H, Cc, U.ctypes.data_as(f32p), lap.ctypes.data_as(f32p)))

    def field_shape(self):
        whc = np.zeros(3, np.int32)
        self._check(self.L.sc_hip_field_shape(self.h, whc.ctypes.data_as(i32p)))
        return int(whc[2]), int(whc[1]), int(whc[0])

    def field_sweep(self, method, sweeps, omega=1.0, sweeps_per_launch=1):
        self._check(self.L.sc_hip_field_sweep(self.h, int(method), int(sweeps), float(omega), int(sweeps_per_launch)))

    def field_residual(self):
        out = np.zeros(2, np.float64)
        self._check(self.L.sc_hip_field_residual(self.h, out.ctypes.data_as(f64p)))
        return float(out[0]), float(out[1])

    def field_solve(self, allow_not_converged=False):
        rc = self.L.sc_hip_field_solve(self.h)
        return self._check(rc, allow=(SC_ERR_NOT_CONVERGED,) if allow_not_converged else ())

    def field_store(self):
        out = np.zeros(self.field_shape(), np.float32)
        self._check(self.L.sc_hip_field_store(self.h, out.ctypes.data_as(f32p), out.size))
        return out

    def field_finish(self, body, ltx, lty):
        """Post-process alone: the field on the device -> clamp, truncate, interleave into `body` (in place)."""
        b = _img(body)
        self._check(self.L.sc_hip_field_finish(self.h, *b, int(ltx), int(lty)))

    def field_lowmode(self):
        """Float-table correction alone on the field on the device (result += correction)."""
        self._check(self.L.sc_hip_field_lowmode(self.h))

    def field_time_sweeps(self, method, launches, sweeps_per_launch=1, omega=1.0) -> float:
        ms = C.c_float(0)
        self._check(self.L.sc_hip_field_time_sweeps(self.h, int(method), int(launches), int(sweeps_per_launch),
                                                    float(omega), C.byref(ms)))
        return float(ms.value)


def _time_cycle0(self, launches: int = 100) -> float:
    ms = C.c_float(0)
    self._check(self.L.sc_hip_time_cycle0(self.h, int(launches), C.byref(ms)))
    return float(ms.value)


Instance.time_cycle0 = _time_cycle0


def _time_cycle0_form(self, form: int, launches: int = 100) -> float:
    ms = C.c_float(0)
    self._check(self.L.sc_hip_time_cycle0_form(self.h, int(form), int(launches), C.byref(ms)))
    return float(ms.value)


Instance.time_cycle0_form = _time_cycle0_form


def _time_coarse_chain(self, reps: int = 50):
    """(ms per pass as plain launches, ms per pass as HIP-graph replays, dependent launches per pass)"""
    a, b, n = C.c_float(0), C.c_float(0), C.c_int(0)
    self._check(self.L.sc_hip_time_coarse_chain(self.h, int(reps), C.byref(a), C.byref(b), C.byref(n)))
    return float(a.value), float(b.value), int(n.value)


Instance.time_coarse_chain = _time_coarse_chain


def _time_tail_phases(self):
    """shader-clock differences between the eleven phase boundaries of one k_mg_tail launch (ten numbers)"""
    buf = (C.c_ulonglong * 11)()
    self._check(self.L.sc_hip_time_tail_phases(self.h, buf))
    return [int(buf[i + 1]) - int(buf[i]) for i in range(10)]


Instance.time_tail_phases = _time_tail_phases


class _Borrowed(Instance):
    """An instance owned by a native pool: same methods, no destroy."""

    def __init__(self, L, handle, gpu_id):          # noqa: D401 - no create call
        self.L, self.h, self.gpu_id = L, handle, gpu_id

    def destroy(self):
        self.h = None


class Pool:
    """The library's native batch driver (csrc/sc_pool.cpp): K instances = K HIP streams on one GPU,
    one C++ worker thread each, jobs pulled from a shared counter."""

    def __init__(self, gpu_id: int = 0, streams: int = 4, group: int = 1, **solver):
        self.L = load()
        self.h = self.L.sc_hip_pool_create(int(gpu_id), int(streams))
        if not self.h:
            raise SeamlessCloneError(SC_ERR_HIP, f"cannot create a pool on GPU {gpu_id} (no CPU fallback)")
        self.gpu_id = gpu_id
        self.instances = [_Borrowed(self.L, self.L.sc_hip_pool_instance(self.h, k), gpu_id)
                          for k in range(self.L.sc_hip_pool_size(self.h))]
        if group != 1 and self.L.sc_hip_pool_set_group(self.h, int(group)) != SC_OK:
            raise SeamlessCloneError(SC_ERR_BAD_ARG, f"bad group size {group}")
        self.group = group
        if solver:
            o = self.instances[0].get_solver()
            for k, v in solver.items():
                setattr(o, k, v)
            rc = self.L.sc_hip_pool_set_solver(self.h, C.byref(o))
            if rc != SC_OK:
                raise SeamlessCloneError(rc, "bad solver options")

    @staticmethod
    def make_jobs(n: int):
        return (BatchJob * n)()

    def set_solver(self, **solver):
        """The same options on every instance of the pool (between batches)."""
        o = self.instances[0].get_solver()
        for k, v in solver.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        rc = self.L.sc_hip_pool_set_solver(self.h, C.byref(o))
        if rc != SC_OK:
            raise SeamlessCloneError(rc, "bad solver options")

    def run(self, jobs, device_resident: bool):
        rc = self.L.sc_hip_pool_run(self.h, jobs, len(jobs), 1 if device_resident else 0)
        if rc != SC_OK:
            raise SeamlessCloneError(rc, "a batch job failed (see jobs[i].rc)")

    def run_host(self, items):
        """items: (face, body, mask, cx, cy) numpy tuples; bodies are blended in place."""
        jobs = self.make_jobs(len(items))
        for j, (face, body, mask, cx, cy) in zip(jobs, items):
            f, b, m = _img(face), _img(body), _img(mask)
            (j.face, j.face_cols, j.face_rows, j.face_step) = f
            (j.body, j.body_cols, j.body_rows, j.body_step) = b
            (j.mask, j.mask_cols, j.mask_rows, j.mask_step) = m
            j.centerX, j.centerY, j.body_restore = int(cx), int(cy), None
        self.run(jobs, device_resident=False)

    def close(self):
        if getattr(self, "h", None):
            self.L.sc_hip_pool_destroy(self.h)
            self.h = None
            self.instances = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_size(W: int, H: int, opts: "SolverOpts | None" = None) -> dict:
    """Host-only: what decides the size class of a W x H ROI (ring included)."""
    out = np.zeros(12, np.int32)
    load().sc_hip_plan_size(int(W), int(H), C.byref(opts) if opts is not None else None, out.ctypes.data_as(i32p))
    keys = ("eligible", "levels", "tail_level", "pad_x", "pad_y", "Kxp", "Kyp", "column_tiles", "row_splits", "direct_nx_ny", "solo_differs", "conditional")
    return dict(zip(keys, out.tolist()))


def plan_groups_pool(sizes, group: int = 0, streams: int = 2, opts: "SolverOpts | None" = None):
    """Host-only: plan_groups as a Pool(streams, group) forms its groups (group 0 = SC_POOL_GROUP_AUTO; jobs largest first)."""
    wh = np.ascontiguousarray(np.asarray(sizes, np.int32).reshape(-1, 2))
    n = wh.shape[0]
    g = np.zeros(n, np.int32); k = np.zeros(n, np.int32)
    rc = load().sc_hip_plan_groups_pool(wh.ctypes.data_as(i32p), n, int(group), int(streams), C.byref(opts) if opts is not None else None,
                                        g.ctypes.data_as(i32p), k.ctypes.data_as(i32p))
    if rc < 0:
        raise SeamlessCloneError(rc, "sc_hip_plan_groups_pool")
    return g.tolist(), k.tolist()


SC_POOL_GROUP_AUTO = 0


def plan_prepare(sizes, opts: "SolverOpts | None" = None) -> int:
    """Host-only: memoise the plans and per-size host tables of these ROI sizes ahead of the calls that will meet them; returns how
    many can join a size class."""
    wh = np.ascontiguousarray(np.asarray(sizes, np.int32).reshape(-1, 2))
    return int(load().sc_hip_plan_prepare(wh.ctypes.data_as(i32p), wh.shape[0], C.byref(opts) if opts is not None else None))


def plan_cache_clear() -> None:
    load().sc_hip_plan_cache_clear()


def plan_groups(sizes, cap: int = 0, opts: "SolverOpts | None" = None):
    """Host-only: how a batch whose members have these ROI sizes [(W, H), ...] (ring included) is partitioned into sets of launches:
    (group_of, kind_of) per member -- kind 0 alone, 1 a same-size group, 2 a size class (csrc/sc_ragged.cpp)."""
    L = load()
    wh = np.ascontiguousarray(np.asarray(sizes, np.int32).reshape(-1, 2))
    n = wh.shape[0]
    g = np.zeros(n, np.int32); k = np.zeros(n, np.int32)
    rc = L.sc_hip_plan_groups(wh.ctypes.data_as(i32p), n, int(cap), C.byref(opts) if opts is not None else None,
                              g.ctypes.data_as(i32p), k.ctypes.data_as(i32p))
    if rc < 0:
        raise SeamlessCloneError(rc, "plan_groups")
    return g.tolist(), k.tolist()


def source_fingerprint() -> str:
    """sha256 (16 hex digits) over the library's sources (csrc/*.hip, *.cpp, *.h and the public header).  Profiles under
    profiles/ record it; bench.py refuses to quote counter figures captured from other sources (traffic_stale)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_PKG, "csrc", "*.h")) + [HEADER_PATH])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def device_count() -> int:
    return int(load().sc_hip_device_count())


def device_pci_bus_id(gpu_id: int) -> str | None:
    """'0000:c1:00.0' of HIP device gpu_id, or None (no such device)."""
    buf = C.create_string_buffer(64)
    if load().sc_hip_device_pci_bus_id(int(gpu_id), buf, 64) != SC_OK:
        return None
    return buf.value.decode() or None
