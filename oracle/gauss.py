"""Oracle: separable Gaussian pre-filter (reference src/gaussian_filter.py:24-94).  Test infrastructure."""
import numpy as np


def prepare_kernel(sigma, ksize):
    """Taps of gaussian_filter.py:47-52: integer offsets arange(-K/2, K/2) truncated to int,
    pdf values stored to float32, then divided by their float32 sum."""
    k = np.zeros(ksize, dtype=np.float32)
    xs = np.arange(-ksize / 2, ksize / 2, 1, dtype=int)
    k[:] = 1.0 / np.sqrt(2.0 * np.pi * sigma ** 2) * np.exp(-xs ** 2 / (2.0 * sigma ** 2))
    k /= np.sum(k)
    return k


def _pass_rows(img, k):
    """One 1-D pass along axis 1 (gaussian_filter.py:62-71 with convolveDirect :24-45).

    Border (:63-66): low side mirrored with the edge sample repeated; high side = the last K//2
    samples copied in their original order (a mirror only when K == 3).  Output p accumulates
    pad[p + K-1 - j] * k[j] for j = 0..K-1 in float32, starting from 0 (:37-40)."""
    K = k.shape[0]
    h = K // 2
    n = img.shape[1]
    # the reference fills the pad from the row itself; for n < h it indexes past the row and raises
    if n < h:
        raise IndexError("image narrower than half the kernel")
    pad = np.empty((img.shape[0], n + 2 * h), dtype=np.float32)
    pad[:, h:h + n] = img
    for j in range(h):
        pad[:, h - 1 - j] = img[:, j]
        pad[:, n + K - 2 - j] = img[:, n - 1 - j]
    acc = np.zeros_like(img, dtype=np.float32)
    for j in range(K):
        # full-convolution index i = p + 2h, sample i - j, then cropped by [h:-h] twice
        acc += pad[:, (2 * h - j):(2 * h - j) + n] * k[j]
    return acc


def filter_with_kernel(img, k):
    """convolveSeparableFilter (gaussian_filter.py:54-85): rows pass over every row, then columns
    pass on the row-filtered image.  Writes into ``img`` and returns it, like the reference."""
    if k.shape[0] % 2 == 0:
        raise ValueError("oracle restates the odd-length path only")
    tmp = _pass_rows(np.asarray(img, dtype=np.float32), k)
    out = _pass_rows(np.ascontiguousarray(tmp.T), k).T
    img[:, :] = out
    return img


def filter_px(img, sigma, ksize):
    """gaussian_filterPx (gaussian_filter.py:92-94)."""
    return filter_with_kernel(img, prepare_kernel(sigma, ksize))


def filter_trunc(img, sigma, truncate):
    """gaussian_filter (gaussian_filter.py:87-90): K = 2*int(truncate*sigma + 0.5) + 1."""
    return filter_with_kernel(img, prepare_kernel(sigma, 2 * int(truncate * sigma + 0.5) + 1))
