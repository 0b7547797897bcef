#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the files profiles/ keeps:
  <tag>_bench_n1.json            the bench line of the un-profiled run
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the same command
  <tag>_pmc_counters.txt         per-kernel, per-launch averages of every --pmc pass
  <tag>_pmc_hbm_traffic.json     HBM bytes per launch of the dominant kernel (FETCH_SIZE doubled on gfx950, per
                                 /opt/skills/guides/MI355X_MICROARCH.md; both counters count KiB)
usage: python tools/profile_summarize.py r1 [dominant-kernel-substring]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dom = sys.argv[2] if len(sys.argv) > 2 else "voigt_nodal_kernel<false>"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench_n1.json"))
stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
assert stats, "no kernel_stats.csv"
shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
stats_p = glob.glob(os.path.join(src, "trace_pipelined", "**", "*_kernel_stats.csv"), recursive=True)
if stats_p:
    shutil.copy(stats_p[0], os.path.join(dst, f"{tag}_bench_kernel_stats_pipelined.csv"))
for extra in ("bench_prof.json", "bench_prof_pipelined.json"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{extra}"))

per = {}
lines = []
for pdir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(pdir):
        continue
    for f in glob.glob(os.path.join(pdir, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
        lines.append(f"# pass {os.path.basename(pdir)}  (rocprofv3 --pmc, its own run; values are averages per launch)")
        for k, v in sorted(agg.items()):
            if k.startswith("__amd") or "at::native" in k:
                continue
            n = max(calls[(k, c)] for c in v)
            row = {c: val / n for c, val in sorted(v.items())}
            per.setdefault(k, {}).update(row)
            lines.append(f"{k[:64]:64s} launches={n:3d} " + " ".join(f"{c}={val:.6g}" for c, val in row.items()))
with open(os.path.join(dst, f"{tag}_pmc_counters.txt"), "w") as fh:
    fh.write("\n".join(lines) + "\n")

hit = [k for k in per if dom in k]
if hit and "FETCH_SIZE" in per[hit[0]] and "WRITE_SIZE" in per[hit[0]]:
    c = per[hit[0]]
    out = {"source": f"profiles/{tag}_pmc_counters.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes)",
           "workload": "C3 n_gpus=1", "kernel": hit[0],
           "fetch_size_kb_per_launch": c["FETCH_SIZE"], "write_size_kb_per_launch": c["WRITE_SIZE"],
           "fetch_correction": 2.0,
           "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0}
    with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
if hit and "SQ_INSTS_VALU" in per[hit[0]]:
    c = per[hit[0]]
    avg_ns = None
    for r in csv.DictReader(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))):
        if dom in r["Name"]:
            avg_ns = float(r["AverageNs"])
    out = {"source": f"profiles/{tag}_pmc_counters.txt (rocprofv3 --pmc, separate passes; per-launch averages)",
           "workload": "C3 n_gpus=1", "kernel": hit[0],
           "sq_insts_valu": c["SQ_INSTS_VALU"], "sq_insts_valu_trans_f32": c.get("SQ_INSTS_VALU_TRANS_F32"),
           "sq_insts_salu": c.get("SQ_INSTS_SALU"), "sq_insts_lds": c.get("SQ_INSTS_LDS"), "sq_insts_smem": c.get("SQ_INSTS_SMEM"),
           "sq_waves": c.get("SQ_WAVES"), "ms_per_launch": avg_ns / 1e6 if avg_ns else None}
    with open(os.path.join(dst, f"{tag}_pmc_valu.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
for extra in ("rehearsal_gloo2.json", "time_c3_thin.txt", "time_overhead.txt", "time_fused_bound.txt", "shard_balance.txt", "time_pipeline.txt",
              "time_c3_uniform.txt", "time_c3_clustered.txt", "tile_spread_uniform.txt", "tile_spread_clustered.txt", "time_dropin.txt",
              "time_xs.txt", "time_xs_round2_kernel.txt", "ubench_crosslayer.txt", "c4_timing.txt", "c5_timing.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{extra}"))
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read()[:1500])
print(open(os.path.join(dst, f"{tag}_pmc_counters.txt")).read())
