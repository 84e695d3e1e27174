#!/usr/bin/env python3
"""Turn one GPU-box collection (gpurun_out/<dir>, made by tools/collect_profiles.sh) into the files under profiles/.

Usage: python tools/make_profiles.py gpurun_out/<dir> [tag]      (tag defaults to r03)
Writes  <tag>_bench_kernel_stats.csv / _by_grid.csv      rocprofv3 --kernel-trace --stats of `python bench.py`
        <tag>_paths_kernel_stats.csv                     the same of `python bench_paths.py`
        <tag>_pmc_hbm_traffic.md, pmc_traffic.json       FETCH_SIZE / WRITE_SIZE passes (gfx950: reads doubled)
        <tag>_pmc_sq.md, pmc_sq.json                     SQ pass: issue fractions per kernel
        <tag>_bench_*.json, <tag>_bench_paths.txt, <tag>_ls_modes.txt   plain runs
"""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
CLOCK_GHZ = 2.4           # peak engine clock: SIMD utilisations below are lower bounds when the chip runs slower
N_SIMD = 256 * 4


def short(name):
    return name.split("(")[0].replace("void ", "")


def level_of(name, dur_ns):
    # both pyramid levels launch k_hs_stream; the full-resolution launches are the long ones
    if "k_hs_stream" in name:
        return "4096x4096" if dur_ns > 100000 else "2048x2048"
    return ""


def trace(path):
    return list(csv.DictReader(open(path)))


def by_kernel(rows, with_level):
    agg = collections.defaultdict(list)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if "Grid_Size_X" in r:
            grid = str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
        else:
            grid = r["Grid_Size"]
        agg[(short(r["Kernel_Name"]), grid, level_of(r["Kernel_Name"], d) if with_level else "")].append(d)
    return agg


def write_stats(rows, out_name, with_level):
    agg = by_kernel(rows, with_level)
    tot = sum(sum(v) for v in agg.values())
    out = [("Name", "Grid", "Split", "Calls", "TotalDurationNs", "AverageNs", "MedianNs", "Percentage", "MinNs", "MaxNs")]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        s = sorted(v)
        out.append((k[0], k[1], k[2], len(v), sum(v), "%.1f" % (sum(v) / len(v)), s[len(s) // 2], "%.2f" % (100 * sum(v) / tot), s[0], s[-1]))
    csv.writer(open(os.path.join(dst, out_name), "w")).writerows(out)
    return out


def pmc(path):
    """{(kernel, grid, level): {counter: [values]}} and durations"""
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name = short(r["Kernel_Name"])
        key = (name, r["Grid_Size"], level_of(name, t))
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(t)
    return d, dur


def traffic_table(fetch_dir, write_dir, title, cmd):
    F, _ = pmc(os.path.join(src, fetch_dir, "p_counter_collection.csv"))
    Wr, _ = pmc(os.path.join(src, write_dir, "p_counter_collection.csv"))
    tab = []
    for k in F:
        f = sum(F[k]["FETCH_SIZE"]) / len(F[k]["FETCH_SIZE"])
        w = sum(Wr[k]["WRITE_SIZE"]) / len(Wr[k]["WRITE_SIZE"]) if k in Wr else 0.0
        tab.append((k[0], k[1], k[2], len(F[k]["FETCH_SIZE"]), f, w, (2 * f + w) * 1024 / 1e6))
    tab.sort(key=lambda r: -r[6] * r[3])
    lines = ["## %s" % title, "", "Command (one pass per counter): `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- %s`" % cmd, "",
             "| kernel | grid | level | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM MB (2*F+W) |", "|---|---|---|---|---|---|---|"]
    for r in tab[:24]:
        lines.append("| %s | %s | %s | %d | %.0f | %.0f | %.1f |" % r)
    return tab, lines


def sq_table(sq_dir, stats_rows, title, cmd):
    S, dur = pmc(os.path.join(src, sq_dir, "p_counter_collection.csv"))
    med = {}
    for k, v in by_kernel(stats_rows, True).items():          # durations of the plain stats run (counters slow a kernel down)
        s = sorted(v)
        med[(k[0], k[1])] = s[len(s) // 2]
    tab = []
    for k, c in S.items():
        m = lambda n: sum(c[n]) / max(1, len(c[n])) if n in c else 0.0
        wc = m("SQ_WAVE_CYCLES")
        if wc <= 0:
            continue
        d_ns = med.get((k[0], k[1]), sorted(dur[k])[len(dur[k]) // 2])
        simd_quads = d_ns * CLOCK_GHZ * N_SIMD / 4.0            # quad-cycles all SIMDs offer during the launch
        tab.append({"kernel": k[0], "grid": k[1], "level": k[2], "launches": len(c["SQ_WAVE_CYCLES"]), "duration_us": d_ns / 1e3,
                    "any_of_wave_cycles": m("SQ_ACTIVE_INST_ANY") / wc, "valu_of_wave_cycles": m("SQ_ACTIVE_INST_VALU") / wc,
                    "lds_of_wave_cycles": m("SQ_ACTIVE_INST_LDS") / wc, "wait_any_of_wave_cycles": m("SQ_WAIT_ANY") / wc,
                    "wait_inst_of_wave_cycles": m("SQ_WAIT_INST_ANY") / wc, "waves_per_simd": wc / simd_quads,
                    "simd_valu_issue": m("SQ_ACTIVE_INST_VALU") / simd_quads, "simd_any_issue": m("SQ_ACTIVE_INST_ANY") / simd_quads})
    tab.sort(key=lambda r: -r["duration_us"] * r["launches"])
    for r in tab:
        r["level"] = r["level"] or ("grid %s" % r["grid"])
    lines = ["## %s" % title, "", "Command: `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
             "SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -- %s`" % cmd, "",
             "SQ counters are quad-cycles summed over wavefronts.  `of wave cycles` = share of a wavefront's resident time; `SIMD issue` = the same "
             "counter over the quad-cycles that 1024 SIMDs offer during the launch at %.1f GHz (a lower bound: the chip clocks down under load); "
             "durations from the run without counters." % CLOCK_GHZ, "",
             "| kernel | level | launches | us | waves/SIMD | VALU of wave cycles | ANY of wave cycles | waiting (parked + issue stall) | SIMD VALU issue | SIMD ANY issue |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for r in tab[:20]:
        lines.append("| %s | %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f + %.2f | %.2f | %.2f |" % (
            r["kernel"], r["level"], r["launches"], r["duration_us"], r["waves_per_simd"], r["valu_of_wave_cycles"], r["any_of_wave_cycles"],
            r["wait_any_of_wave_cycles"], r["wait_inst_of_wave_cycles"], r["simd_valu_issue"], r["simd_any_issue"]))
    return tab, lines


bench_rows = trace(os.path.join(src, "stats", "bench_kernel_trace.csv"))
out = write_stats(bench_rows, tag + "_bench_kernel_stats_by_grid.csv", True)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
paths_rows = trace(os.path.join(src, "paths_stats", "paths_kernel_trace.csv"))
write_stats(paths_rows, tag + "_paths_kernel_stats.csv", False)
for f in ("bench_under_rocprof.json", "bench_default.json", "bench_paths.txt", "bench_with_transfer.json", "bench_batch64.json",
          "bench_batch64_rccl_world1.json", "bench_batch64_one_in_flight.json", "ls_modes.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))

t1, l1 = traffic_table("fetch", "write", "bench.py (headline: HS 2-level pyramid at 4096 x 4096)", "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
t2, l2 = traffic_table("paths_fetch", "paths_write", "bench_paths.py (Liu-Shen, Farneback, dense LK and their pyramids at 2048 x 2048)", "python3 bench_paths.py --reps 2 --cfg3-size 0")
head = ["# HBM traffic per launch from rocprofv3 PMC passes (%s)" % tag, "",
        "Counter values are KiB per dispatch.  On gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads, so the read column is "
        "doubled in `corrected` (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores.", ""]
open(os.path.join(dst, tag + "_pmc_hbm_traffic.md"), "w").write("\n".join(head + l1 + [""] + l2) + "\n")
traffic = {}
for r in t1 + t2:
    name = r[0]
    key = None
    if name.startswith("k_hs_stream<12, true, false"):
        key = "k_hs_stream@" + r[2]
    elif name.startswith("k_hs_stream<16"):
        key = "k_hs_stream_deep@" + r[2]
    elif name.startswith("k_hs_sweep_vec4"):
        key = "k_hs_sweep@4096x4096"
    elif name.startswith("k_ls_stream<6, 2, true, false>") and int(r[1]) > 180000:
        key = "k_ls_stream@2048x2048"
    elif name.startswith("fb_fast::k_fb_window16<5, 0, true, true>") and int(r[1]) > 400000:
        key = "k_fb_window16_flow_matrices@2048x2048"
    elif name.startswith("k_lk_tile"):
        key = "k_lk_tile@1024x1024"
    if key and key not in traffic:
        traffic[key] = r[6] * 1e6
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

s1, m1 = sq_table("sq", bench_rows, "bench.py", "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
s2, m2 = sq_table("paths_sq", paths_rows, "bench_paths.py", "python3 bench_paths.py --reps 2 --cfg3-size 0")
open(os.path.join(dst, tag + "_pmc_sq.md"), "w").write("\n".join(["# Instruction issue per kernel from a rocprofv3 SQ counter pass (%s)" % tag, ""] + m1 + [""] + m2) + "\n")
sq = {}
for r in s1:
    if r["kernel"].startswith("k_hs_stream<12, true, false"):
        sq["k_hs_stream@" + r["level"]] = {k: round(r[k], 4) for k in ("valu_of_wave_cycles", "any_of_wave_cycles", "waves_per_simd", "simd_valu_issue", "simd_any_issue")}
json.dump(sq, open(os.path.join(dst, "pmc_sq.json"), "w"), indent=1)
print(json.dumps(traffic))
print(json.dumps(sq))
for o in out[:8]:
    print(o[0][:48], o[1:])
