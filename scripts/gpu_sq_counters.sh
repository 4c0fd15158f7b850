#!/bin/bash
# SQ occupancy / stall counters per kernel (one PMC pass, --kernel-trace only).  usage: bash scripts/gpu_sq_counters.sh <tag> [bench args]
set -o pipefail
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
python3 - "$OUT" <<'EOF'
import collections, csv, glob, re, sys
out = sys.argv[1]
def norm(name):
    name = re.sub(r"^void\s+", "", name.strip('"')).replace("(anonymous namespace)::", "")
    d = 0
    for i, ch in enumerate(name):
        if ch == "<": d += 1
        elif ch == ">": d -= 1
        elif ch == "(" and d == 0: return name[:i]
    return name
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = norm(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
lines = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    lines.append(f"{k[:48]:48s} n={cnt[k]:4d} wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                 f"active_any {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} active_valu {v.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} "
                 f"valu_insts/wave {v.get('SQ_INSTS_VALU',0)/max(v.get('SQ_WAVES',1),1):9.0f} wavecyc/busy {wc/max(v.get('SQ_BUSY_CYCLES',1),1):6.2f}")
open(out + "/sq_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
EOF
find $OUT/sq -name '*kernel_trace.csv' -delete
