#!/usr/bin/env python3
"""gpurun_out/<tag>_default66/ (tools/profile_default66.sh) -> profiles/<tag>_default66_kernel_stats.csv, _pmc_counters.txt,
_time.txt: the reference's DEFAULT compute_TUD call (66 layers, DVOUT 0.0005 -> 11 M wavenumbers).
usage: python tools/profile_summarize66.py r3"""
import collections, csv, glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"{tag}_default66")
dst = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)[0], os.path.join(dst, f"{tag}_default66_kernel_stats.csv"))
shutil.copy(os.path.join(src, "time.txt"), os.path.join(dst, f"{tag}_default66_time.txt"))
lines = []
for pdir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(pdir):
        continue
    for f in glob.glob(os.path.join(pdir, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(r["Kernel_Name"], r["Counter_Name"])] += 1
        lines.append(f"# pass {os.path.basename(pdir)}  (rocprofv3 --pmc, its own run; averages per launch)")
        for k, v in sorted(agg.items()):
            if k.startswith("__amd") or "at::native" in k:
                continue
            n = max(calls[(k, c)] for c in v)
            lines.append(f"{k[:64]:64s} launches={n:3d} " + " ".join(f"{c}={val / n:.6g}" for c, val in sorted(v.items())))
open(os.path.join(dst, f"{tag}_default66_pmc_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
