"""Horn-Schunck optical flow on the GPU, behind the reference's module interface
(reference src/HornSchunck.py: HSOpticalFlowAlgoAdapter, HS, computeDerivatives)."""
import ctypes as C

import numpy as np

import _ofri


class HSOpticalFlowAlgoAdapter(object):
    """Same constructor and protocol as the reference adapter (HornSchunck.py:29-50): each
    ``compute`` consumes the LAST entry of ``alphas``."""

    def __init__(self, alphas, Niter, provideGenericPyramidalDefaults=True):
        self.provideGenericPyramidalDefaults = provideGenericPyramidalDefaults
        self.alphas = alphas
        self.Niter = Niter

    def compute(self, im1, im2, U, V):
        alpha = self.alphas.pop()
        return HS(im1, im2, alpha, self.Niter, U, V)

    accepts_zero_flow_as_none = True     # compute_device takes U = V = None for a zero initial flow
    can_accumulate = True                # accumulate_device adds the new flow to the driver's running sum itself

    def compute_device(self, im1, im2, U, V, want_error=False):
        """Device-resident form used by the pyramid driver: DeviceArrays in, DeviceArrays out."""
        alpha = self.alphas.pop()
        return hs_device(im1, im2, alpha, self.Niter, U, V, want_error)

    def accumulate_device(self, im1, im2, U, V, Uacc, Vacc):
        """compute_device followed by Uacc += U, Vacc += V (the driver's accumulation,
        GenericPyramidalOpticalFlow.py:413-414) in one call; nothing is returned."""
        alpha = self.alphas.pop()
        ctx = im1.ctx
        rows, cols = im1.shape
        _ofri.check(_ofri.lib().ofri_hs_compute_acc(ctx.handle, im1.ptr, im2.ptr, U.ptr if U is not None else None,
                                                    V.ptr if V is not None else None, Uacc.ptr, Vacc.ptr,
                                                    float(alpha), int(self.Niter), rows, cols))

    can_run_level = True                 # level_device runs warp, pre-filter, step and accumulation of a level

    def level_device(self, im1, im2, Uflow, Vflow, taps, warp):
        """One pyramid level on its resized images (GenericPyramidalOpticalFlow.py:200-201 warp, 374-375
        pre-filter, 406 step from a zero flow, 413-414 accumulation).  warp False: Uflow, Vflow receive the
        flow; warp True: the images are warped by them and the new flow is added to them."""
        alpha = self.alphas.pop()
        ctx = im1.ctx
        rows, cols = im1.shape
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        _ofri.check(_ofri.lib().ofri_hs_level(ctx.handle, im1.ptr, im2.ptr, Uflow.ptr, Vflow.ptr, taps.ctypes.data,
                                              int(taps.size), 1 if warp else 0, float(alpha), int(self.Niter), rows, cols))

    def getAlgoName(self):
        return 'Horn-Schunck'

    def hasGenericPyramidalDefaults(self):
        return self.provideGenericPyramidalDefaults

    def getGenericPyramidalDefaults(self):
        return {'warping': True, 'biLinear': True, 'scaling': True}


def hs_device(frame0, frame1, alpha, Niter, U, V, want_error=True):
    """U = V = None stands for a zero initial flow (no zero planes are made or read on large fields)."""
    ctx = frame0.ctx
    rows, cols = frame0.shape
    Un, Vn = ctx.empty((rows, cols)), ctx.empty((rows, cols))
    err = C.c_double(0.0)
    _ofri.check(_ofri.lib().ofri_hs_compute(ctx.handle, frame0.ptr, frame1.ptr, U.ptr if U is not None else None,
                                            V.ptr if V is not None else None, Un.ptr, Vn.ptr,
                                            float(alpha), int(Niter), rows, cols,
                                            C.byref(err) if want_error else None))
    return Un, Vn, (err.value if want_error else None)


def hs_iterate_device(U, V, fx, fy, ft, alpha, Niter):
    """Niter Jacobi sweeps (reference HS_helper, HornSchunck.py:62-71) in place on DeviceArrays U, V."""
    _ofri.check(_ofri.lib().ofri_hs_iterate(U.ctx.handle, U.ptr, V.ptr, fx.ptr, fy.ptr, ft.ptr, float(alpha),
                                            int(Niter), U.shape[0], U.shape[1]))
    return U, V


def HS(im2, im1, alpha, Niter, U, V):
    """Same (swapped) parameter names as the reference (HornSchunck.py:73): the first argument is the
    frame the adapter received first.  Returns (U, V, total_error) as new float32 arrays."""
    ctx = _ofri.context()
    a, b = ctx.upload(im2), ctx.upload(im1)
    shape = a.shape
    Ud = ctx.upload(np.broadcast_to(np.float32(U), shape))
    Vd = ctx.upload(np.broadcast_to(np.float32(V), shape))
    Un, Vn, err = hs_device(a, b, alpha, Niter, Ud, Vd, True)
    return Un.numpy(), Vn.numpy(), err


def computeDerivatives(im1, im2):
    """fx, fy, ft of the 2x2x2 cube, 'mirror' border (HornSchunck.py:107-127)."""
    ctx = _ofri.context()
    a, b = ctx.upload(im1), ctx.upload(im2)
    fx, fy, ft = ctx.empty(a.shape), ctx.empty(a.shape), ctx.empty(a.shape)
    _ofri.check(_ofri.lib().ofri_hs_derivs(ctx.handle, a.ptr, b.ptr, fx.ptr, fy.ptr, ft.ptr, a.shape[0], a.shape[1]))
    return fx.numpy(), fy.numpy(), ft.numpy()
