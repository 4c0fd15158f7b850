#!/bin/bash
# MFMA / LDS counters per kernel (one PMC pass).  usage: bash scripts/gpu_mfma_counters.sh <tag> [bench args]
#   PROG="scripts/conv3_decoder_time.py 2" bash scripts/gpu_mfma_counters.sh <tag>    profiles that script instead of bench.py
set -o pipefail
TAG=${1:-mfma}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $OUT/pmc -- python3 ${PROG:+$R/}${PROG:-$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing} "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - "$OUT" <<'PY'
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
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[norm(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
lines = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))[:24]:
    busy = max(v.get("SQ_BUSY_CU_CYCLES", 0), 1); wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    lines.append(f"{k[:46]:46s} mfma_busy/cu_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/busy:6.3f}  wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                 f"lds_active {v.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} lds_conflict/wavecyc {v.get('SQ_LDS_BANK_CONFLICT',0)/wc:6.3f}")
open(out + "/mfma_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT/pmc -name '*kernel_trace.csv' -delete
