#!/bin/bash
# kernel timeline of a few bench steps: per-stream busy time, overlap, and the top kernels on the critical (main) stream
set -o pipefail
TAG=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
def norm(n):
    n = re.sub(r"^void\s+", "", n.strip('"')).replace("(anonymous namespace)::", "")
    d = 0
    for i, ch in enumerate(n):
        if ch == "<": d += 1
        elif ch == ">": d -= 1
        elif ch == "(" and d == 0: return n[:i]
    return n
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], norm(r["Kernel_Name"])) for r in rows]
ev.sort()
# last step = after the last adam_kernel but one
adam = [i for i, e in enumerate(ev) if e[3] == "adam_kernel"]
lo = ev[adam[-2]][1] if len(adam) >= 2 else ev[0][0]
hi = ev[adam[-1]][1]
step = [e for e in ev if e[0] >= lo and e[1] <= hi]
span = hi - lo
print(f"last step: {span/1e6:.3f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for s, e, q, n in step: byq[q].append((s, e, n))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"GPU busy (union over queues): {union([(s,e) for s,e,_,_ in step])/1e6:.3f} ms")
for q, iv in byq.items():
    print(f"queue {q}: {len(iv)} kernels, busy {union([(s,e) for s,e,_ in iv])/1e6:.3f} ms, sum {sum(e-s for s,e,_ in iv)/1e6:.3f} ms")
mainq = max(byq, key=lambda q: len(byq[q]))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in byq[mainq]:
    agg[n][0] += 1; agg[n][1] += e - s
print("main-queue kernels:")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {n[:50]:50s} x{c:3d} {t/1e6:7.3f} ms")
PY
find $OUT/trace -name '*kernel_trace.csv' -size +20M -delete
