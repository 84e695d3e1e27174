"""Oracle: dense Lucas-Kanade with large windows.  Test infrastructure.

Restates the reference's OpenCL kernel ``lkDense`` (src/pyrlkDenseLargeW.cl:227-669, GPU branch of the
reductions :111-224) and its host wrapper (src/denseLucasKanade_PyCL.py:33-183) in NumPy float32:
one 8x8 "work-group" of 64 lanes per pixel, every lane owning a 4x4 grid of window samples, the same
lane-tree reduction order, ``mad`` evaluated as a fused multiply-add.

PARITY UNPINNED: the kernel needs PyOpenCL, an OpenCL device with image support and its hardware
bilinear sampler (whose weight precision is implementation defined); none exists where the fixtures
are made and the reference ships no expected output.  Bilinear samples here are exact float32
arithmetic per the OpenCL specification's CLK_FILTER_LINEAR formula with clamp-to-edge addressing.
"""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    """float32 fused multiply-add (product exact in float64, one rounding to float32 up to a
    vanishingly rare double rounding)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def _tree(s):
    """reduce1/2/3 (.cl:111-224) over the last axis (64 lanes, tid = yid*8 + xid)."""
    s = s.copy()
    for h in (32, 16, 8, 4):
        s[..., :h] = s[..., :h] + s[..., h:2 * h]
    return (s[..., 0] + s[..., 1]) + (s[..., 2] + s[..., 3])


def _sample_linear(J, x, y):
    """read_imagef with CLK_FILTER_LINEAR | CLK_ADDRESS_CLAMP_TO_EDGE, unnormalised coordinates."""
    H, W = J.shape
    u = x - f32(0.5); v = y - f32(0.5)
    fu = np.floor(u); fv = np.floor(v)
    a = (u - fu).astype(f32); b = (v - fv).astype(f32)
    i0 = np.clip(fu, -2e9, 2e9).astype(np.int64); j0 = np.clip(fv, -2e9, 2e9).astype(np.int64)
    i1 = i0 + 1; j1 = j0 + 1
    i0 = np.clip(i0, 0, W - 1); i1 = np.clip(i1, 0, W - 1); j0 = np.clip(j0, 0, H - 1); j1 = np.clip(j1, 0, H - 1)
    one = f32(1.0)
    return (((one - a) * (one - b)) * J[j0, i0] + (a * (one - b)) * J[j0, i1] + ((one - a) * b) * J[j1, i0]
            + (a * b) * J[j1, i1]).astype(f32)


def window_weights(win_x, win_y, asym):
    """Per-lane weights of the four sample columns / rows (.cl:339-374): arrays (4, 8)."""
    aL, aR, aT, aB = asym
    lane = np.arange(8)

    def axis(win, a_lo, a_hi):
        w = np.zeros((4, 8), f32)
        w[0] = 1
        if win >= 16:                                    # WSX == 1
            w[1] = 1
            w[2] = (16 + lane < win - a_hi)
            w[3] = (24 + lane < win - a_hi)
        else:
            w[1] = (8 + lane < win - a_hi)
        w[1, 0] = 1 - a_lo                               # "if (xid == 0) wx0 = 1 - asymmetricWindowLeft"
        return w
    return axis(win_x, aL, aR), axis(win_y, aT, aB)


def lk_dense(I, J, U, V, niter=5, half_window=13, asym=(0, 0, 0, 0)):
    """lkDense over every pixel.  Returns new (U, V)."""
    I = np.asarray(I, f32); J = np.asarray(J, f32)
    H, W = I.shape
    win = 2 * half_window + 1
    hw = f32((win - 1) >> 1)
    wx, wy = window_weights(win, win, asym)
    ii, jj = np.mgrid[0:H, 0:W]
    yid, xid = np.mgrid[0:8, 0:8]
    P = H * W
    px = (jj.reshape(P, 1, 1).astype(f32) - hw)          # prevPt.x (integer valued)
    py = (ii.reshape(P, 1, 1).astype(f32) - hw)
    # I patch, gradients and A (.cl:398-490)
    Ip = np.zeros((4, 4, P, 8, 8), f32); Dx = np.zeros_like(Ip); Dy = np.zeros_like(Ip)
    A11 = np.zeros((P, 8, 8), f32); A12 = np.zeros_like(A11); A22 = np.zeros_like(A11)
    bx = px.astype(np.int64); by = py.astype(np.int64)

    def val(ty, tx, dy, dx):
        yy = np.clip(by - 1 + (ty * 8 + yid + 1 + dy), 0, H - 1)
        xx = np.clip(bx - 1 + (tx * 8 + xid + 1 + dx), 0, W - 1)
        return I[yy, xx]
    for ty in range(4):
        for tx in range(4):
            w = f32(wx[tx][xid] * wy[ty][yid]) if (tx and ty) else (wx[tx][xid] if tx else (wy[ty][yid] if ty else f32(1)))
            w = np.broadcast_to(np.asarray(w, f32), (8, 8))
            Ip[ty, tx] = val(ty, tx, 0, 0)
            sx = ((val(ty, tx, -1, 1) + val(ty, tx, 1, 1)) - val(ty, tx, -1, -1)) - val(ty, tx, 1, -1)
            dx = fma(sx, f32(3.0), (val(ty, tx, 0, 1) - val(ty, tx, 0, -1)) * f32(10.0)) * w
            sy = ((val(ty, tx, 1, -1) + val(ty, tx, 1, 1)) - val(ty, tx, -1, -1)) - val(ty, tx, -1, 1)
            dy = fma(sy, f32(3.0), (val(ty, tx, 1, 0) - val(ty, tx, -1, 0)) * f32(10.0)) * w
            Dx[ty, tx], Dy[ty, tx] = dx, dy
            A11 = fma(dx, dx, A11); A12 = fma(dx, dy, A12); A22 = fma(dy, dy, A22)
    A11 = _tree(A11.reshape(P, 64)); A12 = _tree(A12.reshape(P, 64)); A22 = _tree(A22.reshape(P, 64))
    D = fma(A11, A22, -(A12 * A12))
    ok = ~(D < f32(1.192092896e-07))
    with np.errstate(divide='ignore', invalid='ignore'):
        A11 = (A11 / D).astype(f32); A12 = (A12 / D).astype(f32); A22 = (A22 / D).astype(f32)
    u0 = np.asarray(U, f32).reshape(P); v0 = np.asarray(V, f32).reshape(P)
    ppx = (jj.reshape(P).astype(f32) + u0) - hw           # prevPt = (j + u, i + v) - halfWin
    ppy = (ii.reshape(P).astype(f32) + v0) - hw
    locx = np.zeros((4, P, 8, 8), f32); locy = np.zeros((4, P, 8, 8), f32)
    locx[0] = ppx[:, None, None] + (xid.astype(f32) + f32(0.5)); locy[0] = ppy[:, None, None] + (yid.astype(f32) + f32(0.5))
    for t in range(1, 4):
        locx[t] = locx[t - 1] + f32(8); locy[t] = locy[t - 1] + f32(8)
    active = ok.copy()
    for _ in range(int(niter)):
        inb = ~((ppx < -hw) | (ppx >= W) | (ppy < -hw) | (ppy >= H))      # (.cl:517-522)
        active &= inb
        if not active.any():
            break
        idx = np.nonzero(active)[0]
        b1 = np.zeros((idx.size, 8, 8), f32); b2 = np.zeros_like(b1)
        for ty in range(4):
            for tx in range(4):
                if ty and tx:
                    w = f32(wy[ty][yid] * wx[tx][xid])
                elif tx:
                    w = wx[tx][xid]
                elif ty:
                    w = wy[ty][yid]
                else:
                    w = f32(1)
                w = np.broadcast_to(np.asarray(w, f32), (8, 8))
                diff = ((_sample_linear(J, locx[tx][idx], locy[ty][idx]) - Ip[ty, tx][idx]) * w).astype(f32)
                b1 = fma(diff, Dx[ty, tx][idx], b1); b2 = fma(diff, Dy[ty, tx][idx], b2)
        b1 = _tree(b1.reshape(idx.size, 64)); b2 = _tree(b2.reshape(idx.size, 64))
        dxx = (fma(A12[idx], b2, -(A22[idx] * b1)) * f32(32.0)).astype(f32)
        dyy = (fma(A12[idx], b1, -(A11[idx] * b2)) * f32(32.0)).astype(f32)
        ppx[idx] = ppx[idx] + dxx; ppy[idx] = ppy[idx] + dyy
        for t in range(4):
            locx[t][idx] = locx[t][idx] + dxx[:, None, None]; locy[t][idx] = locy[t][idx] + dyy[:, None, None]
        conv = (np.abs(dxx) < f32(0.01)) & (np.abs(dyy) < f32(0.01))
        active[idx[conv]] = False
    Un = np.where(ok, (ppx + hw) - jj.reshape(P).astype(f32), u0).astype(f32)
    Vn = np.where(ok, (ppy + hw) - ii.reshape(P).astype(f32), v0).astype(f32)
    return Un.reshape(H, W), Vn.reshape(H, W)


def vorticity_config(U, V, enabled):
    """evaluateVorticityEnhancement (denseLucasKanade_PyCL.py:75-92) -> [left, right, top, bottom]."""
    if not enabled:
        return [0, 0, 0, 0]
    from scipy.ndimage import convolve
    D = np.array([[0, -1, 0], [0, 0, 0], [0, 1, 0]], f32) * f32(0.5)
    omega = convolve(V, D.T, mode='reflect') - convolve(U, D, mode='reflect')
    if np.mean(omega) < -2e-3:
        return [0, 1, 0, 1]
    if np.mean(omega) > 2e-3:
        return [1, 0, 0, 1]
    return [0, 0, 0, 0]


class DenseLKAdapter(object):
    """denseLucasKanade_PyCl (denseLucasKanade_PyCL.py:33-183) without the OpenCL plumbing."""

    def __init__(self, Niter=5, halfWindow=13, provideGenericPyramidalDefaults=True, enableVorticityEnhancement=False):
        self.Niter, self.halfWindow = Niter, halfWindow
        self.defaults, self.vort = provideGenericPyramidalDefaults, enableVorticityEnhancement

    def compute(self, im1, im2, U, V):
        cfg = vorticity_config(U, V, self.vort)
        Un, Vn = lk_dense(im1, im2, U, V, self.Niter, self.halfWindow, cfg)
        return Un, Vn, True

    def getAlgoName(self):
        return 'OpenCL Dense LK'

    def hasGenericPyramidalDefaults(self):
        return self.defaults

    def getGenericPyramidalDefaults(self):
        return {'warping': False, 'intermediateScaling': True, 'scaling': False}
