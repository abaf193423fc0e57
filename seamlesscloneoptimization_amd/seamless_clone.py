"""Python surface of the library: the class the reference exposes through Boost.Python
(seamlessClone-CUDA/seamlessClone-python-binding/SeamlessClone.h:54-98, implementation
SeamlessClone.cpp:60-143), rebuilt on ctypes over the C ABI.

Same method names and argument meaning:
    loadMatsInSeamlessClone(face, body, mask, centerX, centerY, gpu_id)
    seamlessClone() -> ndarray (H, W, 3) uint8        (body is blended IN PLACE as well: the
                                                       reference wraps the numpy buffer without
                                                       a copy, SeamlessClone.cpp:217, and the
                                                       clone writes into it, seamlessClone_imp.cpp:470)
    sync(), destroy(), mat2py(a), py2mat(a), loadImageInCpp_Demo(path)
"""
from __future__ import annotations

import numpy as np

from . import capi


class SeamlessClone:
    def __init__(self):
        self.instance_ptr = None          # capi.Instance, created lazily like SeamlessClone.cpp:110-113
        self.bSync = False                # SeamlessClone.cpp:63
        self.face = self.body = self.mask = self.blendedMat = None
        self.centerX = self.centerY = 0
        self.gpu_id = 0
        self._opts = {}

    # -- reference surface --------------------------------------------------------------
    def loadMatsInSeamlessClone(self, oface, obody, omask, centerX, centerY, gpu_id):
        self.face = self.py2mat(oface)
        self.body = self.py2mat(obody)
        self.mask = self.py2mat(omask)
        self.centerX, self.centerY, self.gpu_id = int(centerX), int(centerY), int(gpu_id)

    def seamlessClone(self):
        if self.face is None:
            raise RuntimeError("loadMatsInSeamlessClone() has not been called")
        if self.instance_ptr is None:
            self.instance_ptr = capi.Instance(self.gpu_id)
            if self._opts:
                self.instance_ptr.set_solver(**self._opts)
        self.instance_ptr.run(self.face, self.body, self.mask, self.centerX, self.centerY, sync=self.bSync)
        self.blendedMat = self.body       # header copy: aliases the caller's buffer (imp.cpp:470)
        return self.mat2py(self.blendedMat)

    def sync(self):
        if self.instance_ptr is not None:
            self.instance_ptr.sync()

    def destroy(self):
        if self.instance_ptr is not None:
            self.instance_ptr.destroy()
            self.instance_ptr = None

    @staticmethod
    def mat2py(mat):
        """New (rows, cols, channels) array holding a copy (SeamlessClone.cpp:120-143)."""
        a = np.asarray(mat)
        if a.ndim == 2:
            a = a[:, :, None]
        return np.array(a, copy=True, order="C")

    @staticmethod
    def py2mat(o):
        """Zero-copy view of a numpy image (SeamlessClone.cpp:145-226): uint8, HxW or HxWxC,
        rows contiguous."""
        a = np.asarray(o)
        if a.dtype != np.uint8:
            raise TypeError("image data type = %s is not supported" % a.dtype)
        if a.ndim not in (2, 3):
            raise TypeError("dimensionality (=%d) is not supported" % a.ndim)
        if a.strides[-1] != 1:
            a = np.ascontiguousarray(a)   # the reference copies in this case too (needcopy)
        return a

    def loadImageInCpp_Demo(self, imagePath):
        """cv::imread equivalent via PIL (BGR channel order like OpenCV)."""
        from PIL import Image
        rgb = np.asarray(Image.open(imagePath).convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    # -- additions ----------------------------------------------------------------------
    def setSolver(self, **kw):
        """Forwarded to sc_hip_set_solver (method, max_sweeps, tol, omega, ...)."""
        self._opts.update(kw)
        if self.instance_ptr is not None:
            self.instance_ptr.set_solver(**kw)

    def info(self):
        return self.instance_ptr.info() if self.instance_ptr is not None else None


def seamlessClone(src, dst, mask, p, flags=capi.SC_OK + 1, gpu_id=0, **solver):
    """cv2.seamlessClone-shaped convenience: returns a NEW blended image, dst untouched."""
    if flags != 1:
        raise ValueError("only NORMAL_CLONE (1) is implemented, as in the reference")
    out = np.array(dst, np.uint8, copy=True, order="C")
    inst = capi.Instance(gpu_id)
    try:
        if solver:
            inst.set_solver(**solver)
        inst.run(np.ascontiguousarray(src), out, np.ascontiguousarray(mask), int(p[0]), int(p[1]), sync=True)
    finally:
        inst.destroy()
    return out
