"""Liu-Shen physics-based optical flow on the GPU, behind the reference's module interface
(reference src/PhysicsBasedOpticalFlowLiuShen.py: LiuShenOpticalFlowAlgoAdapter,
physicsBasedOpticalFlowLiuShen)."""
import ctypes as C

import numpy as np

import _ofri

MAXNUM = 60      # reference: maxnum (PhysicsBasedOpticalFlowLiuShen.py:88)
TOL = 1e-8       # reference: tol (:89)


class LiuShenOpticalFlowAlgoAdapter(object):
    def __init__(self, alpha):
        self.alpha = alpha

    def compute(self, im1, im2, U, V):
        [resV, resU, error] = physicsBasedOpticalFlowLiuShen(im1, im2, self.alpha, V, U)
        return [resU, resV, error]

    def compute_device(self, im1, im2, U, V, want_error=False):
        rv, ru, err, _ = liushen_device(im1, im2, self.alpha, V, U, want_error)
        return ru, rv, err

    def getAlgoName(self):
        return 'Liu-Shen Physics based OF'

    def hasGenericPyramidalDefaults(self):
        return False


def liushen_device(im1, im2, h, U, V, want_error=True):
    """Arguments as in the reference function: U is the row-direction component."""
    ctx = im1.ctx
    rows, cols = im1.shape
    un, vn = ctx.empty((rows, cols)), ctx.empty((rows, cols))
    err, it = C.c_double(0.0), C.c_int(0)
    _ofri.check(_ofri.lib().ofri_ls_compute(ctx.handle, im1.ptr, im2.ptr, U.ptr, V.ptr, un.ptr, vn.ptr, float(h),
                                            MAXNUM, TOL, rows, cols, C.byref(err) if want_error else None,
                                            C.byref(it) if want_error else None))
    return un, vn, (err.value if want_error else None), (it.value if want_error else None)


def physicsBasedOpticalFlowLiuShen(im1, im2, h, U, V):
    ctx = _ofri.context()
    a, b = ctx.upload(im1), ctx.upload(im2)
    Ud = ctx.upload(np.broadcast_to(np.float32(U), a.shape))
    Vd = ctx.upload(np.broadcast_to(np.float32(V), a.shape))
    un, vn, err, _ = liushen_device(a, b, h, Ud, Vd, True)
    return un.numpy(), vn.numpy(), err
