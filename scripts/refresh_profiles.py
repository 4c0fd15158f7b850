#!/usr/bin/env python3
"""Copy the summaries of a scripts/gpu_measure.sh run (gpurun_out/<tag>/) into profiles/ under per-round names.

usage: python scripts/refresh_profiles.py <tag> <round, e.g. r01> <workload> <batch>
(gpurun_out/ is scratch; profiles/ is what gets committed and judged)"""
import glob
import os
import shutil
import sys

tag, rnd, workload, batch = sys.argv[1:5]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "gpurun_out", tag)
dst = os.path.join(repo, "profiles")
stem = f"{rnd}_{workload}_b{batch}"


def cp(pattern, name):
    files = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    if not files:
        print("missing", pattern)
        return
    shutil.copyfile(files[-1], os.path.join(dst, name))     # newest (gpurun_out/ accumulates across calls)
    print("->", name)


cp("bench.json", f"{stem}_bench.json")
cp("bench_under_rocprofv3.json", f"{stem}_bench_under_rocprofv3.json")
cp("stats/**/*kernel_stats.csv", f"{stem}_rocprofv3_kernel_stats.csv")
cp("pmc_traffic.json", f"{rnd}_pmc_traffic_{workload}_b{batch}.json")
cp("pmc_traffic.txt", f"{rnd}_pmc_traffic_{workload}_b{batch}.txt")
