#!/usr/bin/env python3
"""Turn one GPU-box collection (gpurun_out/<dir>) into the files under profiles/.

The collection is made on the GPU box by (each rocprofv3 run on its own, from /tmp):
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python bench.py > $O/bench_default.json ; python bench_paths.py > $O/bench_paths.txt
Usage: python tools/make_profiles.py gpurun_out/<dir> [tag]      (tag defaults to r01)
"""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")


def level_of(name, dur_ns):
    # both pyramid levels launch k_hs_stream; the full-resolution
    # launches are the long ones
    if "k_hs_stream" in name:
        return "4096x4096 (duration > 120 us)" if dur_ns > 120000 else "2048x2048 (duration <= 120 us)"
    return ""


# ---- kernel durations by kernel, grid and level
rows = list(csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_trace.csv"))))
agg = collections.defaultdict(list)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[(r["Kernel_Name"], r["Grid_Size_X"], level_of(r["Kernel_Name"], d))].append(d)
tot = sum(sum(v) for v in agg.values())
out = [("Name", "Grid_X", "Split", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    out.append((k[0], k[1], k[2], len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.2f" % (100 * sum(v) / tot), min(v), max(v)))
csv.writer(open(os.path.join(dst, tag + "_bench_kernel_stats_by_grid.csv"), "w")).writerows(out)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_default.json", "bench_paths.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))


# ---- HBM traffic per launch from the two PMC passes
def pmc(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[(name, r["Grid_Size"], level_of(name, dur))].append(float(r["Counter_Value"]))
    return d


F = pmc(os.path.join(src, "fetch", "p_counter_collection.csv"))
Wr = pmc(os.path.join(src, "write", "p_counter_collection.csv"))
tab = []
for k in F:
    f = sum(F[k]) / len(F[k])
    w = sum(Wr[k]) / len(Wr[k]) if k in Wr else 0.0
    tab.append((k[0], k[1], k[2], len(F[k]), f, w, (2 * f + w) * 1024 / 1e6))
tab.sort(key=lambda r: -r[6] * r[3])
lines = ["# HBM traffic per launch from rocprofv3 PMC passes (%s)" % tag, "",
         "Command (one pass per counter): `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- "
         "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`", "",
         "Counter values are KiB per dispatch.  On gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads, so the "
         "read column is doubled in `corrected`.", "",
         "| kernel | grid | level | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM MB (2*F+W) |",
         "|---|---|---|---|---|---|---|"]
for r in tab:
    lines.append("| %s | %s | %s | %d | %.0f | %.0f | %.1f |" % r)
open(os.path.join(dst, tag + "_pmc_hbm_traffic.md"), "w").write("\n".join(lines) + "\n")
traffic = {}
for r in tab:
    if r[0].startswith("k_hs_stream<12, true, false>"):
        traffic["k_hs_stream@" + r[2].split(" ")[0]] = r[6] * 1e6
    if r[0].startswith("k_hs_fused<8>"):
        traffic["k_hs_fused@" + ("4096x4096" if int(r[1]) > 500000 else "2048x2048")] = r[6] * 1e6
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic))
for o in out[:6]:
    print(o[0][:48], o[1:])
