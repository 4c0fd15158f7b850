#!/usr/bin/env python3
"""Idle gaps of the main HIP queue within the last step of a kernel trace (scripts/gpu_timeline.sh leaves
gpurun_out/<tag>/trace/**/*kernel_trace.csv): where the critical path waits -- for the side stream, for the host, for nothing.
usage: python scripts/timeline_gaps.py gpurun_out/<tag> [min_gap_us]"""
import collections, csv, glob, re, sys

out = sys.argv[1]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 10e3
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]


def norm(n):
    n = re.sub(r"^void\s+", "", n.strip('"')).replace("(anonymous namespace)::", "")
    d = 0
    for i, ch in enumerate(n):
        if ch == "<": d += 1
        elif ch == ">": d -= 1
        elif ch == "(" and d == 0: return n[:i]
    return n


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], norm(r["Kernel_Name"])) for r in csv.DictReader(open(f)))
adam = [i for i, e in enumerate(ev) if e[3] == "adam_kernel"]
lo, hi = ev[adam[-2]][1], ev[adam[-1]][1]
step = [e for e in ev if e[0] >= lo and e[1] <= hi]
byq = collections.defaultdict(list)
for s, e, q, n in step:
    byq[q].append((s, e, n))
mainq = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
m = sorted(byq[mainq])
gaps = [(s1 - e0, n0, n1, (e0 - lo) / 1e6) for (s0, e0, n0), (s1, e1, n1) in zip(m, m[1:]) if s1 - e0 > 0]
print(f"last step {(hi - lo) / 1e6:.3f} ms; main queue: {len(m)} kernels, idle {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
for q, iv in byq.items():
    if q != mainq:
        iv = sorted(iv)
        print(f"side queue {q}: {len(iv)} kernels, first starts at {(iv[0][0] - lo) / 1e6:.2f} ms, last ends at {(iv[-1][1] - lo) / 1e6:.2f} ms")
for g in sorted(gaps, reverse=True):
    if g[0] >= min_gap:
        print(f"{g[0] / 1e3:9.1f} us  at {g[3]:6.2f} ms  after {g[1][:44]:44s} before {g[2][:44]}")
