"""Farneback polynomial-expansion optical flow on the GPU, behind the reference's module interface
(reference src/Farneback_PyCL.py: class Farneback_PyCL).  The class keeps its historical name and
constructor; ``platformID`` is ignored and ``deviceID`` selects the HIP device."""
import numpy as np

import _ofri
from GaussianKernelBitExact import getGaussianKernelBitExact


def imresize(im, res):
    """Pillow-compatible BILINEAR resize to (width, height) (reference :62-63)."""
    ctx = _ofri.context()
    return resize_bilinear_device(ctx.upload(im), res).numpy()


def resize_bilinear_device(im, size_wh):
    w, h = int(size_wh[0]), int(size_wh[1])
    out = im.ctx.empty((h, w))
    _ofri.check(_ofri.lib().ofri_resize_bilinear(im.ctx.handle, im.ptr, im.shape[0], im.shape[1], out.ptr, h, w))
    return out


_EXPANSION_CONSTS = {}      # (polyN, polySigma) -> FarnebackPrepareGaussian()
_BLUR_TAPS = {}             # (n, sigma) -> float32 taps of getGaussianKernelBitExact


class Farneback_PyCL(object):
    def __init__(self, windowSize=33, Niters=5, polyN=7, polySigma=1.5, useGaussian=True, pyrScale=0.5,
                 pyramidalLevels=1, platformID=0, deviceID=0, provideGenericPyramidalDefaults=True):
        assert pyramidalLevels >= 1, 'Pyramidal levels must be greater or equal than 1'
        self.useDouble = False
        self.windowSize = windowSize
        self.numIters = Niters
        self.polyN = int(polyN)
        self.polySigma = polySigma
        self.useGaussianFilter = useGaussian
        self.pyramidalLevels = pyramidalLevels - 1
        self.fastPyramids = False
        self.pyrScale = pyrScale
        self.provideGenericPyramidalDefaults = provideGenericPyramidalDefaults
        if windowSize & 1 == 0:
            raise Exception('windowSize must be an odd value')
        self.deviceID = int(deviceID)
        self.ctx = _ofri.context(self.deviceID)        # raises if no GPU / library: there is no CPU path

    # ---- host constant tables (reference :124-209) ----------------------------------------------
    def FarnebackPrepareGaussian(self):
        n = self.polyN
        sigma = self.polySigma
        if sigma < 1.19209289550781250000000000000000000e-7:
            sigma = n * 0.3
        xs = range(-n, n + 1)
        g = np.zeros([2 * n + 1], dtype=np.float32)
        xg = np.zeros_like(g)
        xxg = np.zeros_like(g)
        total = np.float64(0.0)
        for x in xs:
            g[x + n] = np.exp(-x * x / (2 * sigma * sigma))
            total += g[x + n]
        inv_total = 1.0 / total
        for x in xs:
            g[x + n] = np.float32(g[x + n] * inv_total)
            xg[x + n] = np.float32(x * g[x + n])
            xxg[x + n] = np.float32(x * x * g[x + n])
        G = np.zeros((6, 6), np.float64)
        for y in xs:
            for x in xs:
                w = g[y + n] * g[x + n]
                G[0, 0] += w
                G[1, 1] += w * x * x
                G[3, 3] += w * x * x * x * x
                G[5, 5] += w * x * x * y * y
        G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
        G[4, 4] = G[3, 3]
        G[3, 4] = G[4, 3] = G[5, 5]
        invG = np.linalg.inv(G)
        return g, xg, xxg, invG[1, 1], invG[0, 3], invG[3, 3], invG[5, 5]

    def setPolynomialExpansionConsts(self):
        n = self.polyN
        key = (int(self.polyN), float(self.polySigma))
        if key not in _EXPANSION_CONSTS:            # the same tables for every call with these parameters (host work of a millisecond)
            _EXPANSION_CONSTS[key] = self.FarnebackPrepareGaussian()
        g, xg, xxg, a, b, c, d = _EXPANSION_CONSTS[key]
        self.matrixIGD = np.array([a, b, c, d], dtype=np.float64)
        self.matrixIG = self.matrixIGD.astype(np.float32)
        self.matrixG = g[n:].reshape([1, n + 1]).copy()
        self.matrixXG = xg[n:].reshape([1, n + 1]).copy()
        self.matrixXXG = xxg[n:].reshape([1, n + 1]).copy()

    def getGaussianKernel(self, n, sigma):
        key = (int(n), float(sigma))
        if key not in _BLUR_TAPS:                   # 28-digit decimal arithmetic per tap: once per (size, sigma)
            _, kernel_bitexact = getGaussianKernelBitExact(n, sigma)
            _BLUR_TAPS[key] = np.float32(kernel_bitexact).reshape([1, n])
        return _BLUR_TAPS[key].copy()

    def setGaussianBlurKernel(self, smoothSize, sigma):
        g = self.getGaussianKernel(smoothSize, sigma)
        m_gKer = np.zeros([1, int(smoothSize / 2) + 1], dtype=np.float32)
        m_gKer[0, :] = g[0, int(smoothSize / 2):]
        self.matrixGKernel = m_gKer

    # ---- kernel wrappers on device arrays ------------------------------------------------------------
    def _blur(self, src, ksh):
        out = src.ctx.empty(src.shape)
        _ofri.check(_ofri.lib().ofri_fb_blur(src.ctx.handle, src.ptr, out.ptr, src.shape[0], src.shape[1],
                                             self.matrixGKernel.ctypes.data, int(ksh)))
        return out

    def _polyexp(self, src):
        h, w = src.shape
        out = src.ctx.empty((5 * h, w))
        _ofri.check(_ofri.lib().ofri_fb_polyexp(src.ctx.handle, src.ptr, out.ptr, h, w, self.polyN,
                                                self.matrixG.ctypes.data, self.matrixXG.ctypes.data,
                                                self.matrixXXG.ctypes.data, self.matrixIG.ctypes.data))
        return out

    def _blur_polyexp(self, src, ksh):
        """gaussianBlur with the taps set last, then polynomialExpansion, without the blurred frame in memory."""
        h, w = src.shape
        out = src.ctx.empty((5 * h, w))
        _ofri.check(_ofri.lib().ofri_fb_blur_polyexp(src.ctx.handle, src.ptr, out.ptr, h, w, self.matrixGKernel.ctypes.data,
                                                     int(ksh), self.polyN, self.matrixG.ctypes.data, self.matrixXG.ctypes.data,
                                                     self.matrixXXG.ctypes.data, self.matrixIG.ctypes.data))
        return out

    def _update_matrices(self, fx, fy, RA, RB, M):
        h, w = fx.shape
        _ofri.check(_ofri.lib().ofri_fb_update_matrices(fx.ctx.handle, fx.ptr, fy.ptr, RA.ptr, RB.ptr, M.ptr, h, w))
        return M

    def _window_flow(self, M, fx, fy, ksh):
        h, w = fx.shape
        _ofri.check(_ofri.lib().ofri_fb_window5_update_flow(fx.ctx.handle, M.ptr, fx.ptr, fy.ptr, h, w,
                                                            self.matrixGKernel.ctypes.data, int(ksh),
                                                            0 if self.useGaussianFilter else 1))

    # ---- NumPy-facing wrappers with the reference's names ----------------------------------------------
    def gaussianBlur(self, src, kSizeHalf, dst):
        dst[:, :] = self._blur(self.ctx.upload(src), int(kSizeHalf)).numpy()
        return dst

    def _plane5(self, fn, src, dst, *a):
        ctx = self.ctx
        s = ctx.upload(src)
        out = ctx.empty(s.shape)
        h = s.shape[0] // 5
        _ofri.check(fn(ctx.handle, s.ptr, out.ptr, h, s.shape[1], *a))
        dst[:, :] = out.numpy()
        return dst

    def gaussianBlur5(self, src, kSizeHalf, dst):
        return self._plane5(_ofri.lib().ofri_fb_blur5, src, dst, self.matrixGKernel.ctypes.data, int(kSizeHalf))

    def boxFilter5(self, src, kSizeHalf, dst):
        return self._plane5(_ofri.lib().ofri_fb_box5, src, dst, int(kSizeHalf))

    def polynomialExpansion(self, src, dst):
        dst[:, :] = self._polyexp(self.ctx.upload(src)).numpy()
        return dst

    def updateFlow(self, M, flowX, flowY):
        ctx = self.ctx
        m = ctx.upload(M)
        fx, fy = ctx.empty(flowX.shape), ctx.empty(flowY.shape)
        _ofri.check(_ofri.lib().ofri_fb_update_flow(ctx.handle, m.ptr, fx.ptr, fy.ptr, flowX.shape[0], flowX.shape[1]))
        flowX[:, :] = fx.numpy()
        flowY[:, :] = fy.numpy()
        return flowX, flowY

    def updateMatrices(self, flowX, flowY, RA, RB, M):
        ctx = self.ctx
        m = ctx.empty(M.shape)
        self._update_matrices(ctx.upload(flowX), ctx.upload(flowY), ctx.upload(RA), ctx.upload(RB), m)
        M[:, :] = m.numpy()
        return M

    def pyrDown(self, pyramid):
        raise Exception('Fast Pyramids not supported yet... Check ocl_pyrDown and ocl_pyrUp, pyrDown, pyrUp in OpenCV')

    # ---- the adapter protocol ------------------------------------------------------------------------------
    def compute(self, im1, im2, U, V):
        assert (self.polyN == 5 or self.polyN == 7)
        assert (im1.shape == im2.shape and self.pyrScale < 1)
        assert (U.shape == im1.shape and V.shape == im1.shape)
        ctx = self.ctx
        fx, fy, err = self.compute_device(ctx.upload(im1), ctx.upload(im2), ctx.upload(U), ctx.upload(V))
        return fx.numpy(), fy.numpy(), err

    def compute_device(self, im1, im2, U, V, want_error=False):
        assert (self.polyN == 5 or self.polyN == 7)
        assert (im1.shape == im2.shape and self.pyrScale < 1)
        assert (U.shape == im1.shape and V.shape == im1.shape)
        lib = _ofri.lib()
        ctx = im1.ctx
        size = im1.shape
        min_size = 32
        scale = 1
        finalNumLevels = 0
        while finalNumLevels < self.pyramidalLevels:        # crop unnecessary internal pyramid levels
            scale *= self.pyrScale
            if size[1] * scale < min_size or size[0] * scale < min_size:
                break
            finalNumLevels += 1
        self.setPolynomialExpansionConsts()
        prevX = prevY = None
        curX = curY = None
        for k in range(finalNumLevels, -1, -1):
            scale = 1.0
            for _ in range(k):
                scale *= self.pyrScale
            sigma = (1.0 / scale - 1.0) * 0.5
            smoothSize = max(int(round(sigma * 5)) | 1, 3)
            width, height = int(round(size[1] * scale)), int(round(size[0] * scale))
            srcX, srcY = (U, V) if prevX is None else (prevX, prevY)
            mult = np.float32(scale) if prevX is None else np.float32(1.0 / self.pyrScale)
            if srcX.shape == (height, width) and mult == 1.0 and self.numIters > 0:
                # the resize is an identity copy (:531-540) and nothing scales it: the incoming flow is read where it is
                # (only the first updateMatrices reads it) and the last iteration's updateFlow writes fresh arrays
                inX, inY = srcX, srcY
                curX, curY = ctx.empty((height, width)), ctx.empty((height, width))
            else:
                curX, curY = resize_bilinear_device(srcX, (width, height)), resize_bilinear_device(srcY, (width, height))
                if mult != 1.0:
                    _ofri.check(lib.ofri_scale_inplace(ctx.handle, curX.ptr, float(mult), curX.size))
                    _ofri.check(lib.ofri_scale_inplace(ctx.handle, curY.ptr, float(mult), curY.size))
                inX, inY = curX, curY
            self.setGaussianBlurKernel(smoothSize, sigma)
            ksh = int(smoothSize / 2)                        # the reference passes smoothSize/2 through cl.cltypes.int
            if (height, width) == tuple(size):               # the resize between blur and expansion (:570-585) is an identity
                RA, RB = self._blur_polyexp(im1, ksh), self._blur_polyexp(im2, ksh)
            else:
                RA = self._polyexp(resize_bilinear_device(self._blur(im1, ksh), (width, height)))
                RB = self._polyexp(resize_bilinear_device(self._blur(im2, ksh), (width, height)))
            M, Mnext = ctx.empty((5 * height, width)), ctx.empty((5 * height, width))
            self._update_matrices(inX, inY, RA, RB, M)
            if self.useGaussianFilter:
                self.setGaussianBlurKernel(self.windowSize, self.windowSize / 2 * 0.3)
            wksh = int(self.windowSize / 2)
            for i in range(self.numIters):
                if i < self.numIters - 1:
                    # window filter + updateFlow + updateMatrices (:447-457) in one call; the displacement of an
                    # iteration that is not the last only feeds the matrices of the next
                    _ofri.check(lib.ofri_fb_window5_flow_matrices(ctx.handle, M.ptr, None, None, RA.ptr, RB.ptr, Mnext.ptr, height, width,
                                                                  self.matrixGKernel.ctypes.data, wksh,
                                                                  0 if self.useGaussianFilter else 1))
                    M, Mnext = Mnext, M
                else:
                    self._window_flow(M, curX, curY, wksh)
            prevX, prevY = curX, curY
        return curX, curY, 'Unknown'

    def getAlgoName(self):
        return 'Farneback CL'

    def hasGenericPyramidalDefaults(self):
        return self.provideGenericPyramidalDefaults

    def getGenericPyramidalDefaults(self):
        return {'warping': False, 'scaling': True}
