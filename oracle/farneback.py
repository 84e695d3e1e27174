"""Oracle: Farneback polynomial-expansion optical flow.  Test infrastructure.

Restates the reference's OpenCL kernels (src/optical_flow_farneback.cl:72-429) and their host
sequencing (src/Farneback_PyCL.py:124-209, 425-604) in NumPy float32, operation by operation in the
order the kernel text writes them, products and sums rounded separately.

PARITY UNPINNED for the kernels: the reference needs PyOpenCL and an OpenCL device to run them, neither
exists in the build environment, and the reference ships no expected outputs.  Only the host constant
tables (prepare_gaussian, bitexact_kernel) are pinned, against tests/golden/farneback_tables.npz.
"""
from decimal import Decimal

import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------- host tables
def prepare_gaussian(n, sigma):
    """FarnebackPrepareGaussian (Farneback_PyCL.py:124-172): g, xg, xxg (float32, length 2n+1) and the
    four entries of the inverse moment matrix (float64)."""
    if sigma < 1.19209289550781250000000000000000000e-7:
        sigma = n * 0.3
    g = np.zeros(2 * n + 1, f32); xg = np.zeros(2 * n + 1, f32); xxg = np.zeros(2 * n + 1, f32)
    s = np.float64(0.0)
    for x in range(-n, n + 1):
        g[x + n] = np.exp(-x * x / (2 * sigma * sigma))
        s += g[x + n]
    s = 1.0 / s
    for x in range(-n, n + 1):
        g[x + n] = f32(g[x + n] * s)
        xg[x + n] = f32(x * g[x + n])
        xxg[x + n] = f32(x * x * g[x + n])
    G = np.zeros((6, 6), np.float64)
    for y in range(-n, n + 1):
        for x in range(-n, n + 1):
            gg = g[y + n] * g[x + n]
            G[0, 0] += gg
            G[1, 1] += gg * x * x
            G[3, 3] += gg * x * x * x * x
            G[5, 5] += gg * x * x * y * y
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    return g, xg, xxg, inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]


_FIXED = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
          7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
          9: [4 / 256, 13 / 256, 30 / 256, 51 / 256, 60 / 256, 51 / 256, 30 / 256, 13 / 256, 4 / 256]}


def bitexact_kernel(n, sigma):
    """getGaussianKernelBitExact (GaussianKernelBitExact.py:55-144) -> float64 taps.  sigma <= 0 with
    n in {1,3,5,7,9} gives fixed tables; otherwise the width is n*0.15+0.35 unless sigma is NEGATIVE
    (the reference's branch at :103-106 ignores a positive sigma)."""
    if sigma <= 0 and n in _FIXED:
        return np.array(_FIXED[n], np.float64)
    sx = Decimal(sigma) if sigma < 0 else Decimal(n) * Decimal('0.15') + Decimal('0.35')
    scale = Decimal('-0.125') / (sx * sx)
    half = int((n - 1) / 2)
    vals = [(Decimal((1 - n + 2 * i) ** 2) * scale).exp() for i in range(half)]
    tot = Decimal(0)
    for v in vals:
        tot += v
    tot = tot * Decimal(2.0) + Decimal('1.0')
    if n % 2 == 0:
        tot += Decimal('1.0')
    mul = Decimal('1.0') / tot
    out = [Decimal(0)] * n
    for i, v in enumerate(vals):
        out[i] = out[n - 1 - i] = v * mul
    out[half] = Decimal('1.0') * mul
    if n % 2 == 0:
        out[half + 1] = out[half]
    return np.array([float(v) for v in out], np.float64)


def blur_taps(n, sigma):
    """setGaussianBlurKernel (Farneback_PyCL.py:203-209): float32 half kernel, centre first."""
    return f32(bitexact_kernel(n, sigma))[int(n / 2):].copy()


# ------------------------------------------------------------------------------- kernels
def _clampi(i, lo, hi):
    return np.minimum(np.maximum(i, lo), hi)


def _low(y, last):                      # idx_row_low / idx_col_low (.cl:135-148)
    return np.abs(y) % (last + 1)


def _high(y, last):                     # idx_row_high / idx_col_high (.cl:140-153)
    return np.abs(last - np.abs(last - y)) % (last + 1)


def poly_expansion(src, n, g, xg, xxg, ig):
    """polynomialExpansion (.cl:72-133).  g, xg, xxg: non-negative halves (length n+1); ig: 4 float32.
    Returns the 5-plane (5H, W) array."""
    src = np.asarray(src, f32)
    H, W = src.shape
    ys = np.arange(H)
    r0 = src * g[0]
    r1 = np.zeros_like(src); r2 = np.zeros_like(src)
    for k in range(1, n + 1):
        t0 = src[_clampi(ys - k, 0, H - 1)]
        t1 = src[_clampi(ys + k, 0, H - 1)]
        r0 = r0 + g[k] * (t0 + t1)
        r1 = r1 + xg[k] * (t1 - t0)
        r2 = r2 + xxg[k] * (t0 + t1)
    xs = np.arange(W)
    b1 = g[0] * r0; b3 = g[0] * r1; b5 = g[0] * r2
    b2 = np.zeros_like(src); b4 = np.zeros_like(src); b6 = np.zeros_like(src)
    for k in range(1, n + 1):
        p, m = _clampi(xs + k, 0, W - 1), _clampi(xs - k, 0, W - 1)
        b1 = b1 + (r0[:, p] + r0[:, m]) * g[k]
        b4 = b4 + (r0[:, p] + r0[:, m]) * xxg[k]
        b2 = b2 + (r0[:, p] - r0[:, m]) * xg[k]
        b3 = b3 + (r1[:, p] + r1[:, m]) * g[k]
        b6 = b6 + (r1[:, p] - r1[:, m]) * xg[k]
        b5 = b5 + (r2[:, p] + r2[:, m]) * g[k]
    return np.concatenate([b3 * ig[0], b2 * ig[0], b1 * ig[1] + b5 * ig[2], b1 * ig[1] + b4 * ig[2], b6 * ig[3]], 0).astype(f32)


def _blur_plane(src, taps, ksh):
    H, W = src.shape
    ys, xs = np.arange(H), np.arange(W)
    row = src * taps[0]
    for j in range(1, ksh + 1):
        row = row + (src[_low(ys - j, H - 1)] + src[_high(ys + j, H - 1)]) * taps[j]
    col = lambda x: _low(_high(x, W - 1), W - 1)       # idx_col (.cl:155-158)
    res = row * taps[0]
    for i in range(1, ksh + 1):
        res = res + (row[:, col(xs - i)] + row[:, col(xs + i)]) * taps[i]
    return res.astype(f32)


def gaussian_blur(src, taps, ksh):
    """gaussianBlur (.cl:160-196): vertical then horizontal pass, reflect-101 borders."""
    return _blur_plane(np.asarray(src, f32), taps, ksh)


def gaussian_blur5(src5, taps, ksh):
    """gaussianBlur5 (.cl:198-253) on a (5H, W) array."""
    H = src5.shape[0] // 5
    return np.concatenate([_blur_plane(src5[k * H:(k + 1) * H], taps, ksh) for k in range(5)], 0)


def box_filter5(src5, ksh):
    """boxFilter5 (.cl:350-406): clamped borders, mean over (2ksh+1)^2."""
    H = src5.shape[0] // 5
    W = src5.shape[1]
    ys, xs = np.arange(H), np.arange(W)
    inv = f32(1.0) / f32((1 + 2 * ksh) * (1 + 2 * ksh))
    out = []
    for k in range(5):
        s = src5[k * H:(k + 1) * H]
        row = s.copy()
        for j in range(1, ksh + 1):
            row = row + (s[_clampi(ys - j, 0, H - 1)] + s[_clampi(ys + j, 0, H - 1)])
        res = row.copy()
        for i in range(1, ksh + 1):
            res = res + (row[:, _clampi(xs - i, 0, W - 1)] + row[:, _clampi(xs + i, 0, W - 1)])
        out.append(res * inv)
    return np.concatenate(out, 0).astype(f32)


_BORDER = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472, 1.0], f32)


def update_matrices(flowx, flowy, R0, R1):
    """updateMatrices (.cl:256-348) -> (5H, W) M."""
    H, W = flowx.shape
    ys, xs = np.mgrid[0:H, 0:W]
    dx = flowx.astype(f32); dy = flowy.astype(f32)
    fx = xs.astype(f32) + dx
    fy = ys.astype(f32) + dy
    x1 = np.floor(fx).astype(np.int64); y1 = np.floor(fy).astype(np.int64)
    fx = fx - x1.astype(f32); fy = fy - y1.astype(f32)
    inside = (x1 >= 0) & (y1 >= 0) & (x1 < W - 1) & (y1 < H - 1)
    xc = np.where(inside, x1, 0); yc = np.where(inside, y1, 0)
    one = f32(1.0)
    a00 = (one - fx) * (one - fy); a01 = fx * (one - fy); a10 = (one - fx) * fy; a11 = fx * fy
    P0 = [R0[k * H:(k + 1) * H] for k in range(5)]
    P1 = [R1[k * H:(k + 1) * H] for k in range(5)]

    def bil(p):
        return a00 * p[yc, xc] + a01 * p[yc, xc + 1] + a10 * p[yc + 1, xc] + a11 * p[yc + 1, xc + 1]
    r2i, r3i = bil(P1[0]), bil(P1[1])
    r4i = (P0[2] + bil(P1[2])) * f32(0.5)
    r5i = (P0[3] + bil(P1[3])) * f32(0.5)
    r6i = (P0[4] + bil(P1[4])) * f32(0.25)
    r2 = np.where(inside, r2i, f32(0)); r3 = np.where(inside, r3i, f32(0))
    r4 = np.where(inside, r4i, P0[2]); r5 = np.where(inside, r5i, P0[3]); r6 = np.where(inside, r6i, P0[4] * f32(0.5))
    r2 = (P0[0] - r2) * f32(0.5)
    r3 = (P0[1] - r3) * f32(0.5)
    r2 = r2 + (r4 * dy + r6 * dx)
    r3 = r3 + (r6 * dy + r5 * dx)
    sc = _BORDER[np.minimum(xs, 5)] * _BORDER[np.minimum(ys, 5)] * _BORDER[np.minimum(W - xs - 1, 5)] * _BORDER[np.minimum(H - ys - 1, 5)]
    r2 = r2 * sc; r3 = r3 * sc; r4 = r4 * sc; r5 = r5 * sc; r6 = r6 * sc
    return np.concatenate([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], 0).astype(f32)


def update_flow(M):
    """updateFlow (.cl:408-429) -> flowx, flowy."""
    H = M.shape[0] // 5
    g11, g12, g22, h1, h2 = (M[k * H:(k + 1) * H] for k in range(5))
    det_inv = f32(1.0) / (g11 * g22 - g12 * g12 + f32(1e-3))
    return ((g11 * h2 - g12 * h1) * det_inv).astype(f32), ((g22 * h1 - g12 * h2) * det_inv).astype(f32)


# ------------------------------------------------------------------------------- host sequencing
def _resize_bilinear(im, wh):
    from PIL import Image
    return np.array(Image.fromarray(im).resize((int(wh[0]), int(wh[1])), Image.BILINEAR))


class FarnebackAdapter(object):
    """Farneback_PyCL (Farneback_PyCL.py:65-616) without the OpenCL plumbing."""

    def __init__(self, windowSize=33, Niters=5, polyN=7, polySigma=1.5, useGaussian=True, pyrScale=0.5,
                 pyramidalLevels=1, provideGenericPyramidalDefaults=True):
        assert pyramidalLevels >= 1
        if windowSize & 1 == 0:
            raise Exception('windowSize must be an odd value')
        self.windowSize, self.numIters, self.polyN, self.polySigma = windowSize, Niters, int(polyN), polySigma
        self.useGaussianFilter, self.pyrScale, self.pyramidalLevels = useGaussian, pyrScale, pyramidalLevels - 1
        self.defaults = provideGenericPyramidalDefaults

    def compute(self, im1, im2, U, V):
        assert self.polyN in (5, 7)
        assert im1.shape == im2.shape and self.pyrScale < 1
        assert U.shape == im1.shape and V.shape == im1.shape
        size = im1.shape
        scale = 1; levels = 0
        while levels < self.pyramidalLevels:
            scale *= self.pyrScale
            if size[1] * scale < 32 or size[0] * scale < 32:
                break
            levels += 1
        n = self.polyN
        g, xg, xxg, i11, i03, i33, i55 = prepare_gaussian(n, self.polySigma)
        ig = np.array([i11, i03, i33, i55], np.float64).astype(f32)
        g, xg, xxg = g[n:].copy(), xg[n:].copy(), xxg[n:].copy()
        prevx = prevy = None
        for k in range(levels, -1, -1):
            scale = 1.0
            for _ in range(k):
                scale *= self.pyrScale
            sigma = (1.0 / scale - 1.0) * 0.5
            smooth = max(int(round(sigma * 5)) | 1, 3)
            width, height = int(round(size[1] * scale)), int(round(size[0] * scale))
            if prevx is None:
                cx = _resize_bilinear(np.asarray(U, f32), (width, height)); cy = _resize_bilinear(np.asarray(V, f32), (width, height))
                cx *= scale; cy *= scale
            else:
                cx = _resize_bilinear(prevx, (width, height)); cy = _resize_bilinear(prevy, (width, height))
                cx *= 1.0 / self.pyrScale; cy *= 1.0 / self.pyrScale
            taps = blur_taps(smooth, sigma)
            ksh = int(smooth / 2)
            RA = poly_expansion(_resize_bilinear(gaussian_blur(im1, taps, ksh), (width, height)), n, g, xg, xxg, ig)
            RB = poly_expansion(_resize_bilinear(gaussian_blur(im2, taps, ksh), (width, height)), n, g, xg, xxg, ig)
            M = update_matrices(cx, cy, RA, RB)
            wtaps = blur_taps(self.windowSize, self.windowSize / 2 * 0.3)
            wksh = int(self.windowSize / 2)
            for i in range(self.numIters):
                M = gaussian_blur5(M, wtaps, wksh) if self.useGaussianFilter else box_filter5(M, wksh)
                cx, cy = update_flow(M)
                if i < self.numIters - 1:
                    M = update_matrices(cx, cy, RA, RB)
            prevx, prevy = cx, cy
        return cx, cy, 'Unknown'

    def getAlgoName(self):
        return 'Farneback CL'

    def hasGenericPyramidalDefaults(self):
        return self.defaults

    def getGenericPyramidalDefaults(self):
        return {'warping': False, 'scaling': True}
