#!/usr/bin/env python3
"""Per-layer A/B of two scripts/profile_ops.py outputs: usage ab_ops.py A.txt B.txt [substring filter]
prints, per (direction, op), the summed kernel time of both runs and which is faster."""
import collections
import re
import sys


def load(path):
    rows = collections.defaultdict(lambda: [0.0, []])
    for l in open(path):
        m = re.match(r'\s*([\d.]+) us (fwd|bwd) (\S+)\s+(.+?)\s+x(\d+)\s+([\d.]+) GB/s\s+([\d.]+) TF', l)
        if m and int(m.group(5)) > 0:
            key = (m.group(2), m.group(3))
            rows[key][0] += float(m.group(1))
            rows[key][1].append(re.sub(r"\s+", " ", m.group(4).strip()))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot_a = tot_b = 0.0
out = []
for key in sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0])[0])):
    if flt and flt not in key[1]:
        continue
    ta, tb = a.get(key, [0.0, []])[0], b.get(key, [0.0, []])[0]
    tot_a += ta
    tot_b += tb
    out.append(f"{key[0]} {key[1][:44]:44s} A {ta:8.1f} us  B {tb:8.1f} us  B/A {tb / ta if ta else 0:5.2f}  A:{'+'.join(a.get(key, [0, []])[1])[:40]:40s} B:{'+'.join(b.get(key, [0, []])[1])[:40]}")
print("\n".join(out))
print(f"total A {tot_a / 1e3:.3f} ms  B {tot_b / 1e3:.3f} ms")
