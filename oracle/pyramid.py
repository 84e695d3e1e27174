"""Oracle: pyramid driver and level-transition glue
(reference src/GenericPyramidalOpticalFlow.py:67-416).  Test infrastructure.

Two flavours of the third-party steps are given: ``*_lib`` calls the same library routine
the reference calls (Pillow ``Image.resize``, SciPy ``RectBivariateSpline``), and
``*_explicit`` spells the arithmetic out so the HIP kernels have a formula to follow.
``tests/test_oracle_*`` checks the two against each other.
"""
import numpy as np

from . import gauss


# --------------------------------------------------------------------------- resize
def resize_lib(im, size_wh):
    """imresize (GenericPyramidalOpticalFlow.py:67-68): Pillow bicubic on a mode-'F' image."""
    from PIL import Image
    return np.array(Image.fromarray(im).resize(tuple(int(s) for s in size_wh), Image.BICUBIC))


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_table(n_in, n_out, support=2.0, kernel=_cubic):
    """Per-output-sample window start, length and normalised float64 weights of Pillow's
    scale-aware (antialiased) resampler: scale = in/out, filter stretched by max(scale, 1)."""
    scale = n_in / n_out
    fs = max(scale, 1.0)
    sup = support * fs
    ksize = int(np.ceil(sup)) * 2 + 1
    start = np.zeros(n_out, np.int32)
    count = np.zeros(n_out, np.int32)
    w = np.zeros((n_out, ksize), np.float64)
    for o in range(n_out):
        c = (o + 0.5) * scale
        lo = max(int(c - sup + 0.5), 0)
        hi = min(int(c + sup + 0.5), n_in)
        ws = np.array([kernel((x + lo - c + 0.5) / fs) for x in range(hi - lo)], np.float64)
        tot = ws.sum()
        if tot != 0.0:
            ws = ws / tot
        start[o], count[o] = lo, hi - lo
        w[o, :hi - lo] = ws
    return start, count, w


def _resample_axis1(im, n_out, support, kernel):
    start, count, w = resample_table(im.shape[1], n_out, support, kernel)
    out = np.empty((im.shape[0], n_out), np.float32)
    imd = im.astype(np.float64)
    for o in range(n_out):
        acc = np.zeros(im.shape[0], np.float64)
        for t in range(count[o]):                      # sequential float64 accumulation
            acc += imd[:, start[o] + t] * w[o, t]
        out[:, o] = acc.astype(np.float32)
    return out


def resize_explicit(im, size_wh, support=2.0, kernel=_cubic):
    """Pillow's two-pass resize on float32: horizontal pass first (float32 intermediate), then
    vertical; a pass whose size does not change is skipped (same-size resize is a copy)."""
    w_out, h_out = int(size_wh[0]), int(size_wh[1])
    cur = np.asarray(im, np.float32)
    if w_out != cur.shape[1]:
        cur = _resample_axis1(cur, w_out, support, kernel)
    if h_out != cur.shape[0]:
        cur = _resample_axis1(np.ascontiguousarray(cur.T), h_out, support, kernel).T
    return np.ascontiguousarray(cur)


# --------------------------------------------------------------------------- flow upsample
def spline_upsample_lib(F, shape_out):
    """Bicubic interpolating spline of the flow evaluated on the finer grid
    (GenericPyramidalOpticalFlow.py:155-162): knots k/n, evaluation points i/N."""
    from scipy.interpolate import RectBivariateSpline
    n0, n1 = F.shape
    N0, N1 = shape_out
    ip = RectBivariateSpline(np.arange(n0) / np.float32(n0), np.arange(n1) / np.float32(n1), F)
    return np.float32(ip(np.arange(N0) / np.float32(N0), np.arange(N1) / np.float32(N1)))


def notaknot_second_derivs(y):
    """Second derivatives (in knot units, h = 1) of the not-a-knot cubic spline through the columns of
    ``y`` (float64, knots along axis 0).  An interpolating FITPACK spline with k=3, s=0 is this spline."""
    n = y.shape[0]
    if n < 4:
        raise ValueError("cubic interpolation needs at least 4 samples")
    # unknowns M_1..M_{n-2}; M_0 = 2 M_1 - M_2 and M_{n-1} = 2 M_{n-2} - M_{n-3} (third derivative
    # continuous across the second and second-to-last knots)
    m = n - 2
    sub = np.ones(m); dia = np.full(m, 4.0); sup = np.ones(m)
    rhs = 6.0 * (y[:-2] - 2.0 * y[1:-1] + y[2:])
    dia[0], sup[0] = 6.0, 0.0
    dia[-1], sub[-1] = 6.0, 0.0
    # Thomas algorithm
    cp = np.zeros(m); dp = np.zeros_like(rhs)
    cp[0] = sup[0] / dia[0]; dp[0] = rhs[0] / dia[0]
    for i in range(1, m):
        den = dia[i] - sub[i] * cp[i - 1]
        cp[i] = sup[i] / den
        dp[i] = (rhs[i] - sub[i] * dp[i - 1]) / den
    M = np.zeros_like(y)
    M[m] = dp[m - 1]
    for i in range(m - 2, -1, -1):
        M[i + 1] = dp[i] - cp[i] * M[i + 2]
    M[0] = 2.0 * M[1] - M[2]
    M[n - 1] = 2.0 * M[n - 2] - M[n - 3]
    return M


def spline_eval_table(n, N):
    """Interval index and fractional offset of evaluation point i/N among knots k/n; points past the
    last knot (n-1)/n are clamped onto it, as FITPACK's evaluator does."""
    i = np.arange(N, dtype=np.int64)
    num = np.minimum(i * n, (n - 1) * N)
    k = np.minimum(num // N, n - 2)
    theta = (num - k * N) / float(N)
    return k.astype(np.int32), theta.astype(np.float64)


def _spline_axis0(y, N):
    n = y.shape[0]
    M = notaknot_second_derivs(y)
    k, th = spline_eval_table(n, N)
    A = (1.0 - th)[:, None]; B = th[:, None]
    return A * y[k] + B * y[k + 1] + ((A ** 3 - A) * M[k] + (B ** 3 - B) * M[k + 1]) / 6.0


def spline_upsample_explicit(F, shape_out):
    """Separable restatement of the tensor-product interpolating spline, float64 throughout,
    rounded to float32 once at the end."""
    t = _spline_axis0(np.asarray(F, np.float64), shape_out[0])
    t = _spline_axis0(np.ascontiguousarray(t.T), shape_out[1]).T
    return np.float32(t)


# --------------------------------------------------------------------------- warp
def warp_bilinear(img, cy, cx):
    """doBiLinearWarping (GenericPyramidalOpticalFlow.py:70-116): sample ``img`` at (cy, cx).

    Nearest index by round-half-even, neighbour on the side of the residual's sign, both clamped to
    the image, weights |residual| - which is ordinary clamp-to-edge bilinear interpolation."""
    iy = np.int32(np.round(cy)); ix = np.int32(np.round(cx))
    dy = cy - iy; dx = cx - ix
    ny = np.where(dy < 0, iy - 1, iy + 1); nx = np.where(dx < 0, ix - 1, ix + 1)
    dy = np.abs(dy); dx = np.abs(dx)
    H, W = img.shape
    iy = np.clip(iy, 0, H - 1); ny = np.clip(ny, 0, H - 1)
    ix = np.clip(ix, 0, W - 1); nx = np.clip(nx, 0, W - 1)
    out = np.zeros(img.shape, np.float32)
    out[:, :] = (1 - dy) * (1 - dx) * img[iy, ix] + (1 - dy) * dx * img[iy, nx] + \
        dy * (1 - dx) * img[ny, ix] + dy * dx * img[ny, nx]
    return out


# --------------------------------------------------------------------------- optical-flow-equation warp
def warp_shift_ofe(im, us, vs):
    """The biLinear=False branch of updateNextPyramidalLevel (GenericPyramidalOpticalFlow.py:189-221).
    Overwrites ``im`` (a float32 array) like the reference and returns it.  ``gaussian_filter`` there is
    the reference's own K-tap filter (gaussian_filter.py:87-90), not SciPy's."""
    H, W = im.shape
    xs, ys = np.meshgrid(np.arange(W, dtype=np.int32), np.arange(H, dtype=np.int32))
    u_sw = np.int32(xs + np.floor(us + 0.5)); v_sw = np.int32(ys + np.floor(vs + 0.5))
    dU = us - np.floor(us + np.float32(0.5)); dV = vs - np.floor(vs + np.float32(0.5))
    im[v_sw, u_sw] = im[ys, xs]
    mask_size = 3
    dU = gauss.filter_trunc(dU, 0.6 * mask_size, 4.0 / 0.6 * mask_size)
    dV = gauss.filter_trunc(dV, 0.6 * mask_size, 4.0 / 0.6 * mask_size)
    tx = (im[0:-1, 1:] * dU[0:-1, 1:] - im[0:-1, 0:-1] * dU[0:-1, 0:-1]) / 1
    ty = (im[1:, 0:-1] * dV[1:, 0:-1] - im[0:-1, 0:-1] * dV[0:-1, 0:-1]) / 1
    im[0:-1, 0:-1] = im[0:-1, 0:-1] - (tx + ty) * 1
    return im


# --------------------------------------------------------------------------- level transition
def next_level(im1, prev_shape, im2, Uacc, Vacc, warping=True, bilinear=True, scale=False,
               upsample=spline_upsample_lib):
    """updateNextPyramidalLevel (GenericPyramidalOpticalFlow.py:118-235).  Returns (im1, im2, Uacc, Vacc,
    U, V) for the new level; with bilinear=False the first image is overwritten, as in the reference."""
    H, W = im1.shape
    if tuple(prev_shape) != (H, W):
        us = upsample(Uacc, (H, W)); vs = upsample(Vacc, (H, W))
    else:
        us, vs = Uacc, Vacc
    if scale:
        us *= np.float32(np.float32(W) / np.float32(prev_shape[1]))
        vs *= np.float32(np.float32(H) / np.float32(prev_shape[0]))
    if not warping:
        return im1, im2, np.zeros((H, W), np.float32), np.zeros((H, W), np.float32), us, vs
    if not bilinear:
        warp_shift_ofe(im1, us, vs)
        return im1, im2, us, vs, np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.int32), np.arange(H, dtype=np.int32))
    w1 = warp_bilinear(im1, np.float32(ys - vs / 2.0), np.float32(xs - us / 2.0))
    w2 = warp_bilinear(im2, np.float32(ys + vs / 2.0), np.float32(xs + us / 2.0))
    return w1, w2, us, vs, np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)


def pyramidal_flow(im1, im2, FILTER, main, levels=1, klevels=1, FILTER_OPT=None, optional=None,
                   warping=True, bilinear=True, intermediate_scaling=True, final_scaling=False,
                   resize=resize_lib, upsample=spline_upsample_lib):
    """genericPyramidalOpticalFlow (GenericPyramidalOpticalFlow.py:238-416)."""
    if main.hasGenericPyramidalDefaults():
        d = main.getGenericPyramidalDefaults() or {}
        if d.get('warping') is not None:
            warping = d.get('warping')
        if d.get('biLinear') is not None:
            bilinear = d.get('biLinear')
        if d.get('intermediateScaling') is not None:
            intermediate_scaling = d.get('intermediateScaling')
        if d.get('scaling') is not None:
            final_scaling = d.get('scaling')
    scale = 1.0 / (2.0 ** (levels - 1))
    work1 = None
    U = V = Uacc = Vacc = None
    for level in range(1, levels + 1):
        lscale = final_scaling if level == levels else intermediate_scaling
        prev = work1
        if scale < 1.0 and level != levels:
            wh = (np.int32(np.round(im1.shape[1] * scale)), np.int32(np.round(im1.shape[0] * scale)))
            n1, n2 = resize(im1, wh), resize(im2, wh)
        elif scale > 1.0:
            raise Exception('Invalid scale level: ' + str(scale))
        else:
            n1, n2 = im1, im2
        if level > 1:
            w1, w2, Uacc, Vacc, U, V = next_level(n1, prev.shape, n2, Uacc, Vacc, warping, bilinear, lscale, upsample)
        else:
            w1, w2 = n1, n2
            U = np.zeros(n1.shape, np.float32); V = np.zeros(n1.shape, np.float32)
            Uacc = np.zeros(n1.shape, np.float32); Vacc = np.zeros(n1.shape, np.float32)
        if FILTER > 1e-3:
            work1 = gauss.filter_px(np.copy(w1), FILTER, 3); work2 = gauss.filter_px(np.copy(w2), FILTER, 3)
        else:
            work1 = np.copy(w1); work2 = w2
        if optional is not None and FILTER_OPT > 1e-3:
            o1 = gauss.filter_px(np.copy(n1), FILTER_OPT, 5); o2 = gauss.filter_px(np.copy(n2), FILTER_OPT, 5)
        elif optional is not None:
            o1 = np.copy(n1); o2 = n2
        for k in range(klevels):
            if k > 0:
                if warping:
                    w1, w2, Uacc, Vacc, U, V = next_level(np.copy(n1), n1.shape, n2, Uacc, Vacc, warping, bilinear, False)
                    if FILTER > 1:
                        work1 = gauss.filter_px(np.copy(w1), FILTER, 3); work2 = gauss.filter_px(np.copy(w2), FILTER, 3)
                    else:
                        work1 = np.copy(w1); work2 = w2
                else:
                    work1, work2, Uacc, Vacc, U, V = next_level(work1, work1.shape, work2, Uacc, Vacc, warping, bilinear, False)
            U, V, _err = main.compute(work1, work2, U, V)
            if optional is not None:
                U, V, _erro = optional.compute(np.copy(o1), np.copy(o2), U, V)
            Uacc += U
            Vacc += V
        scale *= 2
    return Uacc, Vacc
