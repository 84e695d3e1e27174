"""Dense Lucas-Kanade on the GPU, behind the reference's module interface
(reference src/denseLucasKanade_PyCL.py: class denseLucasKanade_PyCl).  ``platformID`` is ignored,
``deviceID`` selects the HIP device."""
import numpy as np

import _ofri


class denseLucasKanade_PyCl(object):
    def __init__(self, platformID=0, deviceID=0, Niter=5, halfWindow=13, provideGenericPyramidalDefaults=True,
                 enableVorticityEnhancement=False):
        self.provideGenericPyramidalDefaults = provideGenericPyramidalDefaults
        self.enableVorticityEnhancement = enableVorticityEnhancement
        self.windowHalfWidth = np.int32(halfWindow)
        self.windowHalfHeight = np.int32(halfWindow)
        self.windowWidth = np.int32(2 * halfWindow + 1)
        self.windowHeight = np.int32(2 * halfWindow + 1)
        self.Niter = Niter
        self.deviceID = int(deviceID)
        self.ctx = _ofri.context(self.deviceID)       # raises if no GPU / library: there is no CPU path

    def evaluateVorticityEnhancement(self, U, V):
        """[left, right, top, bottom] asymmetric-window flags from the sign of the mean vorticity
        (reference :75-92).  The reference CONVOLVES with [-1/2, 0, +1/2], i.e. takes the negated
        central difference, with the border sample repeated."""
        if not self.enableVorticityEnhancement:
            return [0, 0, 0, 0]
        U = np.asarray(U, np.float32)
        V = np.asarray(V, np.float32)
        Vp = np.pad(V, ((0, 0), (1, 1)), mode='symmetric').astype(np.float64)
        Up = np.pad(U, ((1, 1), (0, 0)), mode='symmetric').astype(np.float64)
        Dv = ((Vp[:, :-2] - Vp[:, 2:]) * 0.5).astype(np.float32)
        Du = ((Up[:-2, :] - Up[2:, :]) * 0.5).astype(np.float32)
        omega = Dv - Du
        if np.mean(omega) < -2e-3:
            return [0, 1, 0, 1]
        elif np.mean(omega) > 2e-3:
            return [1, 0, 0, 1]
        return [0, 0, 0, 0]

    def calcPatchSize(self, windowWidth, windowHeight):
        if windowWidth > 32 and windowWidth > 2 * windowHeight:
            blockX, blockY = 32, 8
        else:
            blockX, blockY = 16, 16
        return np.int32((windowWidth + blockX - 1) / blockX), np.int32((windowHeight + blockY - 1) / blockY)

    def compute(self, im1, im2, U, V):
        ctx = self.ctx
        cfg = self.evaluateVorticityEnhancement(U, V)
        Un, Vn, calcErr = self._run(ctx.upload(im1), ctx.upload(im2), ctx.upload(U), ctx.upload(V), cfg)
        return Un.numpy(), Vn.numpy(), calcErr

    def compute_device(self, im1, im2, U, V, want_error=False):
        cfg = [0, 0, 0, 0]
        if self.enableVorticityEnhancement:
            cfg = self.evaluateVorticityEnhancement(U.numpy(), V.numpy())
        return self._run(im1, im2, U, V, cfg)

    def _run(self, im1, im2, U, V, cfg):
        rows, cols = im1.shape
        Un, Vn = im1.ctx.empty((rows, cols)), im1.ctx.empty((rows, cols))
        _ofri.check(_ofri.lib().ofri_lk_compute(im1.ctx.handle, im1.ptr, im2.ptr, U.ptr, V.ptr, Un.ptr, Vn.ptr, rows, cols,
                                                int(self.Niter), int(self.windowHalfWidth), int(cfg[0]), int(cfg[1]),
                                                int(cfg[2]), int(cfg[3])))
        return Un, Vn, True       # the reference returns its calcErr flag (True at level 0) as the "error"

    def getAlgoName(self):
        return 'OpenCL Dense LK'

    def hasGenericPyramidalDefaults(self):
        return self.provideGenericPyramidalDefaults

    def getGenericPyramidalDefaults(self):
        return {'warping': False, 'intermediateScaling': True, 'scaling': False}
