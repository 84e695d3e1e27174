"""Synthetic PIV frame pairs shaped like the reference's bundled Poiseuille test images
(examples/testImages/Bits08/Ni06/parabolic01_{0,1}.tif: 8-bit, ~6 particles per 16x16 px window,
~3 px particle images, parabolic u(y) in -x with a 4 px peak, v = 0).  Host-side utility for the
benchmark and the tests; not part of the compute path."""
import numpy as np


def poiseuille_u(H, peak=-4.0):
    y = np.arange(H, dtype=np.float64)
    c = (H - 1) / 2.0
    return peak * (1.0 - ((y - c) / c) ** 2)


def synth_piv(H, W, seed=0, density=6.0 / 256.0, sigma_p=0.75, peak=-4.0):
    """Return (frame0, frame1) as uint8 arrays.  Particles are rendered at x -/+ u(y)/2."""
    rng = np.random.default_rng(seed)
    pad = 8
    n = int(density * (H + 2 * pad) * (W + 2 * pad))
    py = rng.uniform(-pad, H + pad, n)
    px = rng.uniform(-pad, W + pad, n)
    amp = rng.uniform(50.0, 255.0, n)
    c = (H - 1) / 2.0
    u = peak * (1.0 - ((py - c) / c) ** 2)
    frames = []
    r = 3
    offs = np.arange(-r, r + 1)
    for sgn in (-0.5, +0.5):
        img = np.zeros((H, W), np.float32)
        x = px + sgn * u
        ix = np.rint(x).astype(np.int64)
        iy = np.rint(py).astype(np.int64)
        for dy in offs:
            yy = iy + dy
            wy = np.exp(-((yy - py) ** 2) / (2 * sigma_p ** 2))
            oky = (yy >= 0) & (yy < H)
            for dx in offs:
                xx = ix + dx
                ok = oky & (xx >= 0) & (xx < W)
                w = amp[ok] * wy[ok] * np.exp(-((xx[ok] - x[ok]) ** 2) / (2 * sigma_p ** 2))
                np.add.at(img, (yy[ok], xx[ok]), w.astype(np.float32))
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    return frames[0], frames[1]
