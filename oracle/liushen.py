"""Oracle: Liu-Shen physics-based refinement
(reference src/PhysicsBasedOpticalFlowLiuShen.py:33-158).  Test infrastructure."""
import numpy as np
from scipy.ndimage import convolve


def _flip(k):
    return np.flipud(np.fliplr(k))


# stencils of PhysicsBasedOpticalFlowLiuShen.py:110-121, already mirrored as the reference does, so
# that ``convolve`` applies the un-mirrored matrix as a correlation
D = _flip(np.array([[0, -1, 0], [0, 0, 0], [0, 1, 0]], np.float32) / 2)
M = _flip(np.array([[1, 0, -1], [0, 0, 0], [-1, 0, 1]], np.float32) / 4)
F = _flip(np.array([[0, 1, 0], [0, 0, 0], [0, 1, 0]], np.float32))
D2 = _flip(np.array([[0, 1, 0], [0, -2, 0], [0, 1, 0]], np.float32))
H8 = _flip(np.array([[1, 1, 1], [1, 0, 1], [1, 1, 1]], np.float32))


def inverse_matrix(im, h):
    """generate_invmatrix (PhysicsBasedOpticalFlowLiuShen.py:47-73) with dx = 1."""
    h = np.float32(h)
    cmtx = convolve(np.ones(im.shape, np.float32), H8, mode='constant')
    A11 = im * (convolve(im, D2, mode='nearest') - 2 * im) - h * cmtx
    A22 = im * (convolve(im, D2.transpose(), mode='nearest') - 2 * im) - h * cmtx
    A12 = im * convolve(im, M, mode='nearest')
    det = A11 * A22 - A12 * A12
    return A22 / det, -A12 / det, A11 / det


def coefficient_fields(i1, i2):
    """IIx, IIy, II, Ixt, Iyt (PhysicsBasedOpticalFlowLiuShen.py:124-128), f = 0, dx = dt = 1."""
    IIx = i1 * convolve(i1, D, mode='nearest')
    IIy = i1 * convolve(i1, D.transpose(), mode='nearest')
    II = i1 * i1
    Ixt = i1 * convolve((i2 - i1) / 1 - 0, D, mode='nearest')
    Iyt = i1 * convolve((i2 - i1) / 1 - 0, D.transpose(), mode='nearest')
    return IIx, IIy, II, Ixt, Iyt


def iterate_once(u, v, h, IIx, IIy, II, Ixt, Iyt, B11, B12, B22):
    """One fixed-point sweep (PhysicsBasedOpticalFlowLiuShen.py:142-150, helper :75-80)."""
    Dt, Ft = D.transpose(), F.transpose()
    bu = 2 * IIx * convolve(u, D, mode='nearest') + IIx * convolve(v, Dt, mode='nearest') + \
        IIy * convolve(v, D, mode='nearest') + II * convolve(u, F, mode='nearest') + \
        II * convolve(v, M, mode='nearest') + h * convolve(u, H8, mode='constant') + Ixt
    bv = IIy * convolve(u, D, mode='nearest') + IIx * convolve(u, Dt, mode='nearest') + \
        2 * IIy * convolve(v, Dt, mode='nearest') + II * convolve(u, M, mode='nearest') + \
        II * convolve(v, Ft, mode='nearest') + h * convolve(v, H8, mode='constant') + Iyt
    un = -(B11 * bu + B12 * bv)
    vn = -(B12 * bu + B22 * bv)
    err = (np.linalg.norm(un - u) + np.linalg.norm(vn - v)) / (u.shape[0] * u.shape[1])
    return un, vn, err


def liushen(im1, im2, h, U, V, maxnum=60, tol=1e-8, return_iters=False):
    """physicsBasedOpticalFlowLiuShen (PhysicsBasedOpticalFlowLiuShen.py:82-158).  ``U`` here is the
    row-direction component (the adapter swaps the arguments)."""
    i1 = im1 / np.max(im1)
    i2 = im2 / np.max(im2)
    IIx, IIy, II, Ixt, Iyt = coefficient_fields(i1, i2)
    u = np.float32(U); v = np.float32(V)
    B11, B12, B22 = inverse_matrix(i1, h)
    k = 0; total = 1e8; err = 0
    while total > tol and k < maxnum:
        u, v, total = iterate_once(u, v, h, IIx, IIy, II, Ixt, Iyt, B11, B12, B22)
        err = total
        k += 1
    return (u, v, err, k) if return_iters else (u, v, err)


class LiuShenAdapter(object):
    """LiuShenOpticalFlowAlgoAdapter (PhysicsBasedOpticalFlowLiuShen.py:33-45)."""

    def __init__(self, alpha):
        self.alpha = alpha

    def compute(self, im1, im2, U, V):
        rv, ru, err = liushen(im1, im2, self.alpha, V, U)
        return [ru, rv, err]

    def getAlgoName(self):
        return 'Liu-Shen Physics based OF'

    def hasGenericPyramidalDefaults(self):
        return False
