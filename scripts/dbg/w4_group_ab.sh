#!/bin/bash
# SSDSEG_W4_GROUP (channel tiles per group of the F(4x4) kernel's work order) A/B: parity, isolated kernel time, HBM fetch per launch (PMC),
# full-step ms.  usage (gpurun, repo root): bash scripts/dbg/w4_group_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/w4_group
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 1 2 4 8; do
  export SSDSEG_W4_GROUP=$g
  echo "== group $g"
  timeout -k 10 120 python3 $R/scripts/conv3_decoder_time.py 5 2>&1 | grep -E "wino4_kernel|^y:|^dx:" | awk '{print "   ", $0}'
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$g -- python3 $R/scripts/conv3_decoder_time.py 2 > $OUT/pmc_$g.log 2>&1 || { echo pmc failed; exit 1; }
  python3 - $OUT/pmc_$g <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "wino4_kernel" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
v = [sum(x) for x in agg.values()]
# FETCH_SIZE counts 32-byte units on gfx950 after the guide's correction used by scripts/pmc_summary.py: see that script for the factor
print("    FETCH_SIZE raw per launch (sum over XCDs): min %.0f max %.0f n=%d" % (min(v), max(v), len(v)))
PY
done
unset SSDSEG_W4_GROUP
cd $R
bash scripts/gpu_knob_sweep.sh SSDSEG_W4_GROUP=1 SSDSEG_W4_GROUP=2 SSDSEG_W4_GROUP=4
