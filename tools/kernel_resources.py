#!/usr/bin/env python3
"""Registers, scratch memory, LDS and occupancy of every kernel in csrc/ as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  Scratch > 0 means register spills or a dynamically indexed local array.
    python tools/kernel_resources.py [--scratch-only]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "opticalflow-ri_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -fno-slp-vectorize".split()
only = "--scratch-only" in sys.argv
for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"],
                       cwd=SRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    cur = {}
    for line in r.stdout.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
        if "lds" in cur and "name" in cur:
            if not only or cur.get("scratch", 0) > 0:
                name = subprocess.run(["c++filt", cur["name"]], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
                print("%-24s vgpr %3d agpr %3d scratch %4d occ %d lds %6d  %s" % (os.path.basename(f), cur.get("vgpr", -1), cur.get("agpr", 0),
                      cur.get("scratch", 0), cur.get("occ", 0), cur["lds"], name.split("(")[0][:110]))
            cur = {}
