"""Separable Gaussian pre-filter on the GPU, behind the reference's module interface
(reference src/gaussian_filter.py: prepareGaussianKernel, convolveSeparableFilter, gaussian_filter,
gaussian_filterPx).  As in the reference the filter works in place on its argument and returns it."""
import ctypes as C

import numpy as np

import _ofri


def prepareGaussianKernel(sigma, kernelSizePx):
    """Taps as the reference builds them (gaussian_filter.py:47-52): float32, normalised."""
    k = np.zeros(int(kernelSizePx), dtype=np.float32)
    _ofri.check(_ofri.lib().ofri_gauss_taps(float(sigma), int(kernelSizePx), k.ctypes.data))
    return k


def filter_device(src, kernel, dst=None):
    """Filter DeviceArray ``src`` with float32 taps ``kernel`` into ``dst`` (new array if None;
    pass ``dst=src`` for in place)."""
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    if dst is None:
        dst = src.ctx.empty(src.shape)
    _ofri.check(_ofri.lib().ofri_gauss_sep(src.ctx.handle, src.ptr, dst.ptr, src.shape[0], src.shape[1],
                                           kernel.ctypes.data, kernel.shape[0]))
    return dst


def convolveSeparableFilter(kernel, image):
    if image.ndim != 2:
        raise ValueError("image must be 2-D")
    ctx = _ofri.context()
    d = ctx.upload(image)
    filter_device(d, kernel, d)
    image[:, :] = d.numpy()
    return image


def gaussian_filter(image, sigma, truncate):
    kernelSizePx = 2 * int(truncate * sigma + 0.5) + 1
    return convolveSeparableFilter(prepareGaussianKernel(sigma, kernelSizePx), image)


def gaussian_filterPx(image, sigma, kernelSizePx):
    return convolveSeparableFilter(prepareGaussianKernel(sigma, kernelSizePx), image)
