#!/bin/bash
# bench.py (full step) under one environment switch at a time, two runs each: ms/step per setting.
# usage: bash scripts/gpu_knob_sweep.sh "VAR=value" "VAR2=value" ...   (the unmodified default is always measured first and last)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/knobs
one() {
  for i in 1 2; do
    env "$@" timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/knobs/last.json 2>/dev/null || { echo "$* FAILED"; return; }
    python3 -c "
import json,sys;d=json.load(open('$R/gpurun_out/knobs/last.json'));print('%-40s %8.2f img/s %7.3f ms/step' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"
  done
}
one DEFAULT=1
for s in "$@"; do one $s; done
one DEFAULT=1
