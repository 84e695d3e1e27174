"""Oracle: Horn-Schunck (reference src/HornSchunck.py:29-127).  Test infrastructure."""
import numpy as np
from scipy.ndimage import convolve

# averaging kernel of HornSchunck.py:87-89, stored in float32 like the reference
AVG = np.array([[1 / 12, 1 / 6, 1 / 12], [1 / 6, 0, 1 / 6], [1 / 12, 1 / 6, 1 / 12]], np.float32)


def derivatives(first, second):
    """computeDerivatives(im1=first, im2=second) (HornSchunck.py:107-127), 'mirror' border.

    NB the adapter chain hands frame 1 as ``first`` and frame 0 as ``second``
    (compute -> HS(im2, im1, ...) -> computeDerivatives(im1, im2), HornSchunck.py:37,73,84)."""
    kx = np.array([[-1, 1], [-1, 1]], np.float32) * .25
    ky = np.array([[-1, -1], [1, 1]], np.float32) * .25
    kt = np.ones((2, 2), np.float32) * .25
    fx = convolve(first, kx, mode='mirror') + convolve(second, kx, mode='mirror')
    fy = convolve(first, ky, mode='mirror') + convolve(second, ky, mode='mirror')
    ft = convolve(second, kt, mode='mirror') + convolve(first, -kt, mode='mirror')
    return fx, fy, ft


def jacobi(alpha, niter, U, V, fx, fy, ft):
    """HS_helper + HS_helper2 (HornSchunck.py:52-71): niter Jacobi sweeps in float32."""
    a2 = np.float32(np.float32(alpha) ** 2)
    den = a2 + fx ** 2 + fy ** 2
    for _ in range(int(niter)):
        ua = convolve(U, AVG, mode='mirror')
        va = convolve(V, AVG, mode='mirror')
        der = (fx * ua + fy * va + ft) / den
        U = ua - fx * der
        V = va - fy * der
    return U, V


def hs(frame0, frame1, alpha, niter, U, V):
    """HS as reached from the adapter (HornSchunck.py:35-37,73-105): returns (U, V, error)."""
    fx, fy, ft = derivatives(frame1, frame0)
    U0, V0 = np.float32(U), np.float32(V)
    Un, Vn = jacobi(alpha, niter, U0, V0, fx, fy, ft)
    err = (np.linalg.norm(Un - U) + np.linalg.norm(Vn - V)) / (frame0.shape[0] * frame0.shape[1])
    return Un, Vn, err


class HSAdapter(object):
    """HSOpticalFlowAlgoAdapter (HornSchunck.py:29-50)."""

    def __init__(self, alphas, niter, defaults=True):
        self.alphas, self.niter, self.defaults = alphas, niter, defaults

    def compute(self, im1, im2, U, V):
        return hs(im1, im2, self.alphas.pop(), self.niter, U, V)

    def getAlgoName(self):
        return 'Horn-Schunck'

    def hasGenericPyramidalDefaults(self):
        return self.defaults

    def getGenericPyramidalDefaults(self):
        return {'warping': True, 'biLinear': True, 'scaling': True}
