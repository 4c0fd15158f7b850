#!/bin/bash
# One measurement pass on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel-trace stats of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE cannot
#   share a pass; PMC is never combined with other trace domains).  Everything lands under gpurun_out/<tag>/.
# usage: bash scripts/gpu_measure.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-measure}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 || exit 4
python3 $R/scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
# keep only the summaries small enough to merge back
find $OUT/stats -name '*kernel_trace.csv' -delete
find $OUT/pmc_fetch $OUT/pmc_write -name '*kernel_trace.csv' -delete
python3 - "$OUT" <<'EOF'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
print("img/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d.get("kernel_ms_per_step"))
for k in d.get("kernels_survey_step", []):
    print(" ", k)
EOF
