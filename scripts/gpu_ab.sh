#!/bin/bash
# quick A/B on the GPU box: bench both workloads a few times, print img/s and the heaviest kernels
# usage: bash scripts/gpu_ab.sh <tag> [reps]
TAG=${1:-ab}; REPS=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for w in backbone full; do
  for i in $(seq $REPS); do
    timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $w > $R/gpurun_out/${TAG}_$w.json 2> $R/gpurun_out/${TAG}_$w.err || { echo "bench $w failed"; tail -5 $R/gpurun_out/${TAG}_$w.err; exit 1; }
    python3 - $R/gpurun_out/${TAG}_$w.json $w $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "run", sys.argv[3], "img/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d.get("kernel_ms_per_step"), "dominant", d["roofline"]["kernel"], d["roofline"]["frac"], "isolated", d.get("roofline_isolated", {}).get("frac"))
if sys.argv[3] == "1":
    for k in d.get("kernels_survey_step", [])[:10]:
        print("    %-48s x%-3d %8.3f ms %7.0f GB/s %6.1f TF" % (k["kernel"], k["launches"], k["ms_per_step"], k["GB/s"], k["TFLOP/s"]))
PY
  done
done
