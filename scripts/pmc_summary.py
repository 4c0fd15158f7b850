#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

usage: python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json>

Collected with (one pass per counter, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline

Units and the gfx950 correction (same guide, "HBM" section): the counters are in KiB-like units of 1024 B; FETCH_SIZE
tallies each 128-B request of a wide (16 B / lane) streaming read as 64 B, so it is DOUBLED here (all the streaming kernels
of this library read float4 per lane); WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are counted
as traffic, so `hbm_bytes_per_launch` is an upper bound on what really reached the HBM stacks.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def normalise(name: str) -> str:
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):          # cut the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect(directory: str, counter: str):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = normalise(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        fb = 2.0 * 1024.0 * vf / nf if nf else 0.0      # gfx950: x2 (see module docstring)
        wb = 1024.0 * vw / nw if nw else 0.0
        rows[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                   "hbm_bytes_per_launch": fb + wb, "raw_FETCH_SIZE_per_launch": vf / nf if nf else 0.0,
                   "raw_WRITE_SIZE_per_launch": vw / nw if nw else 0.0}
    json.dump({"unit": "bytes", "correction": "FETCH_SIZE x 1024 x 2 (gfx950 wide-load tally), WRITE_SIZE x 1024", "kernels": rows},
              open(out, "w"), indent=1)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_fetch_pass"])[:25]:
        print(f"{k[:64]:64s} n={v['launches_fetch_pass']:4d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
