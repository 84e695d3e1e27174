#!/usr/bin/env python
"""Timing-and-range report over the ten method configurations of the reference's benchmark
(reference benchmark_of_methods.py:142-331), run on the GPU engine of this repository.

    python methods_report.py                      # bundled 512x512 pair (committed fixture)
    python methods_report.py --pair a.tif b.tif   # any 8/16-bit image pair (16-bit rescaled to 0..255)
    python methods_report.py --synth 2048         # synthetic Poiseuille PIV pair
    python methods_report.py --out DIR --mat --plots

Writes ``DIR/benchmark_summary.txt`` with the reference's columns (method, seconds, U min/max, V min/max);
``--mat`` adds one ``.mat`` per method in the layout of the example scripts, ``--plots`` the v-component
colour map + quiver figure per method and the bar chart of execution times.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "opticalflow-ri_amd"))


def method_table():
    """(name, family, filter_sigma, pyr_levels, Liu-Shen as the main adapter) -- benchmark_of_methods.py:143-148,
    197-201, 251-255.  The 'LiuSE_*' rows of the reference run Liu-Shen(0.1) alone through the pyramid."""
    rows = []
    for fam, fs in (("HS", None), ("LK", 2.0), ("FB", 0.0)):
        if fam == "HS":
            rows += [("HS_Fs0_0", fam, 0.0, 1, False), ("HS_Fs3_4", fam, 3.4, 1, False)]
            fs = 3.4
        else:
            rows += [("%s_Fs%s" % (fam, ("%.1f" % fs).replace(".", "_")), fam, fs, 1, False)]
        tag = "%s_Fs%s_PyrLvls2" % (fam, ("%.1f" % fs).replace(".", "_"))
        rows += [(tag, fam, fs, 2, False), ("LiuSE_" + tag, fam, fs, 2, True)]
    return rows


def make_adapter(family, levels, liu_shen):
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    from Farneback_PyCL import Farneback_PyCL
    if liu_shen:
        return LiuShenOpticalFlowAlgoAdapter(0.1)
    if family == "HS":
        return HSOpticalFlowAlgoAdapter(alphas=[1.0] * levels, Niter=100, provideGenericPyramidalDefaults=True)
    if family == "LK":
        return denseLucasKanade_PyCl(halfWindow=13, Niter=5)
    return Farneback_PyCL(windowSize=33, Niters=5, polyN=7, polySigma=1.5)


def load_pair(args):
    if args.pair:
        from piv_io import imread                            # tifffile, Pillow or the built-in TIFF reader, whichever is there
        a, b = (imread(p).astype(np.float32) for p in args.pair)
    elif args.synth:
        from piv_synth import synth_piv
        a, b = synth_piv(args.synth, args.synth, seed=0)
        a, b = a.astype(np.float32), b.astype(np.float32)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "pair_parabolic01.npz"))
        a, b = g["im0"].astype(np.float32), g["im1"].astype(np.float32)
    if a.max() > 255:                                        # benchmark_of_methods.py:134-137
        a = (a / 65535.0 * 255.0).astype(np.float32)
        b = (b / 65535.0 * 255.0).astype(np.float32)
    if args.crop:
        a, b = a[:args.crop, :args.crop].copy(), b[:args.crop, :args.crop].copy()
    return a, b


def write_mat(U, V, path):
    import scipy.io
    vel = dict(u=U, v=V, iaWidth=1, iaHeight=1, margins=dict(top=0, left=0, bottom=0, right=0))
    par = dict(overlapFactor=1.0, imageHeight=U.shape[0], imageWidth=U.shape[1])
    scipy.io.savemat(path, mdict=dict(velocities=vel, parameters=par))


def write_figure(U, V, title, path, dpi):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    lim = max(abs(np.percentile(V, 1)), abs(np.percentile(V, 99))) or 1.0
    step = max(1, min(U.shape) // 13)
    fig, (left, right) = plt.subplots(1, 2, figsize=(16, 8))
    m = left.imshow(V, cmap="jet", vmin=-lim, vmax=lim)
    left.set_title(title + " - Vertical Velocity (v)")
    fig.colorbar(m, ax=left, label="Pixels/frame")
    yy, xx = np.mgrid[0:U.shape[0]:step, 0:U.shape[1]:step]
    us, vs = U[::step, ::step], V[::step, ::step]
    mag = np.hypot(us, vs)
    q = right.quiver(xx, yy, us, vs, mag, scale=50, scale_units="inches", cmap="jet", clim=[0, np.percentile(mag, 95)])
    fig.colorbar(q, ax=right, label="Magnitude (pixels/frame)")
    right.set_title(title + " - Vector Field")
    right.set_xlim(0, U.shape[1]); right.set_ylim(U.shape[0], 0)
    right.grid(True, alpha=0.3)
    fig.tight_layout()
    fig.savefig(path, dpi=dpi)
    plt.close(fig)


def run(args):
    from GenericPyramidalOpticalFlowWrapper import GenericPyramidalOpticalFlowWrapper
    import _ofri
    os.makedirs(args.out, exist_ok=True)
    a, b = load_pair(args)
    ctx = _ofri.context()
    results = []
    for name, family, fs, levels, liu in method_table():
        if args.only and name not in args.only:
            continue
        flow = GenericPyramidalOpticalFlowWrapper(make_adapter(family, levels, liu), filter_sigma=fs, pyr_levels=levels)
        if args.warm:                                         # first-use costs (tables, pool growth) kept out of the timing
            GenericPyramidalOpticalFlowWrapper(make_adapter(family, levels, liu), filter_sigma=fs,
                                               pyr_levels=levels).calculateFlow(a, b)
        ctx.sync()
        t0 = time.time()
        U, V = flow.calculateFlow(a, b)                       # host arrays in, host arrays out, like the reference's timing
        dt = time.time() - t0
        results.append((name, dt, U, V))
        print("%-30s %8.4f s   U %.2f/%.2f   V %.2f/%.2f" % (name, dt, U.min(), U.max(), V.min(), V.max()))
        if args.mat:
            write_mat(U, V, os.path.join(args.out, name + ".mat"))
        if args.plots:
            write_figure(U, V, name, os.path.join(args.out, name + ".png"), args.dpi)
    with open(os.path.join(args.out, "benchmark_summary.txt"), "w") as f:
        f.write("Optical Flow Methods Benchmark Summary\n=====================================\n\n")
        f.write("%-30s %-10s %-20s %-20s\n" % ("Method", "Time (s)", "U min/max", "V min/max"))
        f.write("-" * 80 + "\n")
        for name, dt, U, V in results:
            f.write("%-30s %-10.2f %-20s %-20s\n" % (name, dt, "%.2f/%.2f" % (U.min(), U.max()), "%.2f/%.2f" % (V.min(), V.max())))
        f.write("\nimage %d x %d; times in ms: %s\n" % (a.shape[0], a.shape[1],
                                                       ", ".join("%s %.1f" % (n, 1e3 * t) for n, t, _, _ in results)))
    if args.plots and results:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig = plt.figure(figsize=(12, 6))
        plt.bar([r[0] for r in results], [r[1] for r in results])
        plt.xlabel("Method"); plt.ylabel("Execution Time (seconds)")
        plt.title("Optical Flow Methods - Execution Time Comparison")
        plt.xticks(rotation=45, ha="right")
        plt.tight_layout()
        fig.savefig(os.path.join(args.out, "execution_time_comparison.png"), dpi=args.dpi)
        plt.close(fig)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--pair", nargs=2, metavar=("FIRST", "SECOND"))
    ap.add_argument("--synth", type=int, default=0, metavar="N", help="synthetic N x N pair")
    ap.add_argument("--crop", type=int, default=0, metavar="N", help="use the top-left N x N crop")
    ap.add_argument("--out", default="benchmark_results")
    ap.add_argument("--only", nargs="*", help="method names to run (default: all ten)")
    ap.add_argument("--mat", action="store_true")
    ap.add_argument("--plots", action="store_true")
    ap.add_argument("--dpi", type=int, default=100)
    ap.add_argument("--no-warm", dest="warm", action="store_false", help="time the very first call of each method")
    return run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
