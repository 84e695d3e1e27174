#!/usr/bin/env python3
"""Loops of the compiled kernels that load from memory and wait for every load before going on (s_waitcnt vmcnt(0)
inside the loop): each round of such a loop costs a full memory latency.
    python tools/isa_load_loops.py [file.hip ...]      (default: every file of csrc/)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "opticalflow-ri_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -fno-slp-vectorize --cuda-device-only -S".split()
files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(SRC, "*.hip")))
for f in files:
    out = "/tmp/_isa_%s.s" % os.path.basename(f)
    if os.path.exists(out):
        os.remove(out)
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", out], cwd=SRC, stderr=subprocess.DEVNULL, check=True)
    txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip().split("(")[0][:90]
        body = m.group(2).split("\n")
        labels = {}
        for i, l in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        found = []
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = body[labels[mm.group(1)]:i]
                nl = sum(1 for s in seg if re.search(r"\b(global|buffer|flat|scratch)_load", s))
                nw = sum(1 for s in seg if "vmcnt(0)" in s)
                if nl > 0 and nw > 0:
                    found.append("%d instrs, %d loads, %d full waits" % (len(seg), nl, nw))
        if found:
            print("%-22s %-90s %s" % (os.path.basename(f), name, "; ".join(found)))
