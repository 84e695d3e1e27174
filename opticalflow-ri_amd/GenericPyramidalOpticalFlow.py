"""Coarse-to-fine pyramid driver with every per-level step on the GPU, behind the reference's module
interface (reference src/GenericPyramidalOpticalFlow.py: imresize, doBiLinearWarping,
updateNextPyramidalLevel, genericPyramidalOpticalFlow).

Images and flow stay in device memory across a whole call; an adapter that offers
``compute_device(im1, im2, U, V)`` (all adapters of this package) is fed device arrays, any other
object implementing the reference's adapter protocol is fed NumPy arrays.
"""
import os

import numpy as np

import _ofri
from gaussian_filter import prepareGaussianKernel, filter_device

VERBOSE = bool(int(os.environ.get("OFRI_VERBOSE", "0")))


def _say(*a):
    if VERBOSE:
        print(*a)


# ------------------------------------------------------------------------------- device steps
def resize_device(im, size_wh):
    """Pillow-compatible antialiased bicubic resize to (width, height)."""
    w, h = int(size_wh[0]), int(size_wh[1])
    out = im.ctx.empty((h, w))
    _ofri.check(_ofri.lib().ofri_resize_bicubic(im.ctx.handle, im.ptr, im.shape[0], im.shape[1], out.ptr, h, w))
    return out


def resize_pair_device(im1, im2, size_wh):
    """Both frames of a level in one launch (reference :336-343)."""
    w, h = int(size_wh[0]), int(size_wh[1])
    o1, o2 = im1.ctx.empty((h, w)), im1.ctx.empty((h, w))
    _ofri.check(_ofri.lib().ofri_resize_bicubic_pair(im1.ctx.handle, im1.ptr, im2.ptr, im1.shape[0], im1.shape[1], o1.ptr, o2.ptr, h, w))
    return o1, o2


def spline_upsample_device(F, shape, post_scale=1.0):
    out = F.ctx.empty(shape)
    _ofri.check(_ofri.lib().ofri_spline_upsample(F.ctx.handle, F.ptr, F.shape[0], F.shape[1], out.ptr,
                                                 int(shape[0]), int(shape[1]), float(post_scale)))
    return out


def warp_flow_device(im, U, V, half_sign):
    out = im.ctx.empty(im.shape)
    _ofri.check(_ofri.lib().ofri_warp_flow(im.ctx.handle, im.ptr, U.ptr, V.ptr, float(half_sign), out.ptr,
                                           im.shape[0], im.shape[1]))
    return out


def shift_warp_device(im, us, vs):
    """The biLinear=False warp (reference :189-221): integer shift of every pixel, then the optical-flow
    equation on the Gaussian-smoothed fractional displacements.  Returns a new array (the reference
    overwrites its argument; callers that rely on that take the result in its place).  Raises IndexError
    where the reference's fancy-index assignment does (a destination outside [-n, n))."""
    mask_size = 3
    sigma, truncate = 0.6 * mask_size, 4.0 / 0.6 * mask_size                 # :210-212
    taps = prepareGaussianKernel(sigma, 2 * int(truncate * sigma + 0.5) + 1)   # gaussian_filter.py:87-89
    out = im.ctx.empty(im.shape)
    _ofri.check(_ofri.lib().ofri_warp_shift_ofe(im.ctx.handle, im.ptr, us.ptr, vs.ptr, out.ptr, im.shape[0], im.shape[1],
                                                taps.ctypes.data, int(taps.size)))
    return out


def upsample_flow_device(Uacc, Vacc, shape, prev_shape, scale):
    """The flow so far on the next level's grid (reference :152-162, scaling :147-150 and :163-175)."""
    ctx = Uacc.ctx
    H, W = shape
    sx = np.float32(np.float32(W) / np.float32(prev_shape[1])) if scale else np.float32(1.0)
    sy = np.float32(np.float32(H) / np.float32(prev_shape[0])) if scale else np.float32(1.0)
    if tuple(prev_shape) != (H, W):
        us, vs = ctx.empty((H, W)), ctx.empty((H, W))        # both components in one pair of launches
        _ofri.check(_ofri.lib().ofri_spline_upsample_pair(ctx.handle, Uacc.ptr, Vacc.ptr, Uacc.shape[0], Uacc.shape[1], us.ptr, vs.ptr,
                                                          int(H), int(W), float(sx), float(sy)))
    else:
        us, vs = Uacc, Vacc
        if scale:
            _say('Scaling  ')
            if sx != 1.0:
                _ofri.check(_ofri.lib().ofri_scale_inplace(ctx.handle, us.ptr, float(sx), us.size))
            if sy != 1.0:
                _ofri.check(_ofri.lib().ofri_scale_inplace(ctx.handle, vs.ptr, float(sy), vs.size))
    return us, vs


def next_level_device(im1, prev_shape, im2, Uacc, Vacc, warping=True, biLinear=True, scale=False):
    """Level transition (reference updateNextPyramidalLevel, :118-235) on device arrays.  After a warp the
    new level starts from a zero flow (:224-225), returned as U = V = None (see _compute).  With
    warping and not biLinear the reference overwrites its first image; here the first returned array is
    that image and the argument is left alone."""
    ctx = im1.ctx
    H, W = im1.shape
    us, vs = upsample_flow_device(Uacc, Vacc, (H, W), prev_shape, scale)
    if not warping:
        return im1, im2, ctx.zeros((H, W)), ctx.zeros((H, W)), us, vs
    if not biLinear:
        _say('Warping: Liu-Shen')
        return shift_warp_device(im1, us, vs), im2, us, vs, None, None
    _say('Warping: BiLinear')
    w1, w2 = ctx.empty((H, W)), ctx.empty((H, W))
    _ofri.check(_ofri.lib().ofri_warp_pair(ctx.handle, im1.ptr, im2.ptr, us.ptr, vs.ptr, w1.ptr, w2.ptr, H, W))
    return w1, w2, us, vs, None, None


def _compute(adapter, im1, im2, U, V):
    """Run one adapter step on device arrays; falls back to the NumPy protocol for foreign adapters."""
    ctx = im1.ctx
    if U is None and not getattr(adapter, "accepts_zero_flow_as_none", False):
        U, V = ctx.zeros(im1.shape), ctx.zeros(im1.shape)
    if hasattr(adapter, "compute_device"):
        Un, Vn, err = adapter.compute_device(im1, im2, U, V, want_error=VERBOSE)
        return Un, Vn, err
    r = adapter.compute(im1.numpy(), im2.numpy(), U.numpy(), V.numpy())
    return ctx.upload(r[0]), ctx.upload(r[1]), r[2]


def pyramidal_flow_device(im1, im2, FILTER, mainOFlowAlgoAdapter, pyramidalLevels=1, kLevels=1, FILTER_OPT=None,
                          optionalOFlowAlgoAdapter=None, warping=True, biLinear=True,
                          pyramidalIntermediateScaling=True, pyramidalScaling=False, overwritten=None):
    """genericPyramidalOpticalFlow on DeviceArrays; returns (Uaccum, Vaccum) DeviceArrays without
    synchronising the stream.  ``overwritten``: optional dict; with warping and not biLinear the reference
    overwrites the caller's first image at the last level (:207,221,347), and the image it would hold
    afterwards is stored under 'im1'."""
    ctx = im1.ctx
    lib = _ofri.lib()
    main, opt = mainOFlowAlgoAdapter, optionalOFlowAlgoAdapter
    if main.hasGenericPyramidalDefaults():
        d = main.getGenericPyramidalDefaults()
        if d is not None:
            for key, label in (('warping', 'warping'), ('biLinear', 'biLinear'),
                               ('intermediateScaling', 'intermediatePyramidalScaling'), ('scaling', 'pyramidalScaling')):
                if d.get(key) is not None:
                    _say('Using algorithm ' + main.getAlgoName() + ' default value for ' + label + ' parameter:', d.get(key))
            warping = d.get('warping') if d.get('warping') is not None else warping
            biLinear = d.get('biLinear') if d.get('biLinear') is not None else biLinear
            if d.get('intermediateScaling') is not None:
                pyramidalIntermediateScaling = d.get('intermediateScaling')
            if d.get('scaling') is not None:
                pyramidalScaling = d.get('scaling')

    taps = prepareGaussianKernel(FILTER, 3) if FILTER > 1e-3 else None
    taps_opt = prepareGaussianKernel(FILTER_OPT, 5) if (opt is not None and FILTER_OPT > 1e-3) else None
    # Horn-Schunck alone, one pass per level, bilinear warp: the adapter takes the level's images as resized and
    # runs warp, pre-filter, its step and the accumulation itself (fused into fewer passes over memory)
    whole_level = (opt is None and kLevels == 1 and warping and biLinear and taps is not None and not VERBOSE
                   and getattr(main, "can_run_level", False))
    scale = 1.0 / (2.0 ** (pyramidalLevels - 1))
    work1 = work2 = None
    U = V = Uacc = Vacc = None
    rows, cols = im1.shape
    for level in range(1, pyramidalLevels + 1):
        lscale = pyramidalScaling if level == pyramidalLevels else pyramidalIntermediateScaling
        prev = work1
        if scale < 1.0 and level != pyramidalLevels:
            wh = (int(np.int32(np.round(cols * scale))), int(np.int32(np.round(rows * scale))))
            n1, n2 = resize_pair_device(im1, im2, wh)
        elif scale > 1.0:
            raise Exception('Invalid scale level: ' + str(scale))
        else:
            n1, n2 = im1, im2
        if whole_level:
            if level > 1:
                Uacc, Vacc = upsample_flow_device(Uacc, Vacc, n1.shape, prev.shape, lscale)
            else:
                Uacc, Vacc = ctx.empty(n1.shape), ctx.empty(n1.shape)
            main.level_device(n1, n2, Uacc, Vacc, taps, warp=level > 1)
            work1 = n1                                # the next level asks for this level's shape only
            scale *= 2
            continue
        if level > 1:
            w1, w2, Uacc, Vacc, U, V = next_level_device(n1, prev.shape, n2, Uacc, Vacc, warping, biLinear, lscale)
            if warping and not biLinear:
                n1 = w1              # the reference's level image now is the warped one (:207,221)
                if level == pyramidalLevels and overwritten is not None:
                    overwritten['im1'] = w1
        else:
            w1, w2 = n1, n2
            U = V = None                         # zero flow (:361-362)
            Uacc = Vacc = None                   # zero so far: the first accumulation adopts U, V instead of adding
        if taps is not None:
            work1, work2 = filter_device(w1, taps), filter_device(w2, taps)
        else:
            work1, work2 = w1, w2                # nothing downstream writes into the level images
        if opt is not None:
            if taps_opt is not None:
                o1, o2 = filter_device(n1, taps_opt), filter_device(n2, taps_opt)
            else:
                o1, o2 = n1, n2
        for k in range(kLevels):
            _say('Level=', level, ' kIter=', k)
            if k > 0:
                if warping:
                    w1, w2, Uacc, Vacc, U, V = next_level_device(n1, n1.shape, n2, Uacc, Vacc, warping, biLinear, False)
                    if FILTER > 1:
                        work1, work2 = filter_device(w1, taps), filter_device(w2, taps)
                    else:
                        work1, work2 = w1, w2
                else:
                    work1, work2, Uacc, Vacc, U, V = next_level_device(work1, work1.shape, work2, Uacc, Vacc,
                                                                       warping, biLinear, False)
            if opt is None and Uacc is not None and not VERBOSE and getattr(main, "can_accumulate", False):
                # the adapter adds its increment to the running sum itself (in its last launch's stores)
                main.accumulate_device(work1, work2, U, V, Uacc, Vacc)
                continue
            U, V, err = _compute(main, work1, work2, U, V)
            _say(main.getAlgoName() + ' estimated error for image registration: ' + str(err))
            if opt is not None:
                U, V, erro = _compute(opt, o1, o2, U, V)
                _say(opt.getAlgoName() + ' estimated error for image registration: ' + str(erro))
            if Uacc is None:
                Uacc, Vacc = U, V                # 0 + U compares equal to U (only a -0 would turn into +0)
            else:
                _ofri.check(lib.ofri_add_inplace(ctx.handle, Uacc.ptr, U.ptr, Uacc.size))
                _ofri.check(lib.ofri_add_inplace(ctx.handle, Vacc.ptr, V.ptr, Vacc.size))
        scale *= 2
    return Uacc, Vacc


# ------------------------------------------------------------------------------- reference API
def imresize(im, res):
    ctx = _ofri.context()
    return resize_device(ctx.upload(im), res).numpy()


def doBiLinearWarping(img, coordsY, coordsX, order=1, mode='nearest'):
    ctx = _ofri.context()
    d = ctx.upload(img)
    cy, cx = ctx.upload(coordsY), ctx.upload(coordsX)
    out = ctx.empty(d.shape)
    _ofri.check(_ofri.lib().ofri_warp_coords(ctx.handle, d.ptr, cy.ptr, cx.ptr, out.ptr, d.shape[0], d.shape[1]))
    return out.numpy()


def updateNextPyramidalLevel(im1IterNext, im1IterPrev, im2IterNext, Uaccum, Vaccum, U, V, warping=True,
                             biLinear=True, scale=False):
    ctx = _ofri.context()
    r = next_level_device(ctx.upload(im1IterNext), im1IterPrev.shape, ctx.upload(im2IterNext), ctx.upload(Uaccum),
                          ctx.upload(Vaccum), warping, biLinear, scale)
    out = [a.numpy() if a is not None else np.zeros(r[0].shape, np.float32) for a in r]
    if warping and not biLinear and isinstance(im1IterNext, np.ndarray):
        im1IterNext[...] = out[0]            # the reference warps its first argument in place and returns it
        out[0] = im1IterNext
    return tuple(out)


def genericPyramidalOpticalFlow(im1, im2, FILTER, mainOFlowAlgoAdapter, pyramidalLevels=1, kLevels=1,
                                FILTER_OPT=None, optionalOFlowAlgoAdapter=None, warping=True, biLinear=True,
                                pyramidalIntermediateScaling=True, pyramidalScaling=False):
    """Same signature and return value as the reference (GenericPyramidalOpticalFlow.py:238-416):
    NumPy images in, (Uaccum, Vaccum) float32 NumPy arrays out."""
    ctx = _ofri.context()
    changed = {}
    Ua, Va = pyramidal_flow_device(ctx.upload(im1), ctx.upload(im2), FILTER, mainOFlowAlgoAdapter, int(pyramidalLevels),
                                   int(kLevels), FILTER_OPT, optionalOFlowAlgoAdapter, warping, biLinear,
                                   pyramidalIntermediateScaling, pyramidalScaling, changed)
    if 'im1' in changed and isinstance(im1, np.ndarray):
        im1[...] = changed['im1'].numpy()    # as the reference leaves the caller's first image (biLinear=False)
    return Ua.numpy(), Va.numpy()
