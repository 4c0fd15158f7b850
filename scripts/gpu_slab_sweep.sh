#!/bin/bash
# bench.py under different caps of the weight-gradient partial-slab traffic (csrc/gemm.hip::slab_fraction)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/slab
mkdir -p $OUT
for f in 0.5 0.25 0.125 1.0; do
  SSDSEG_WGRAD_SLAB_FRAC=$f timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-kernels > $OUT/bench_$f.json 2> $OUT/bench_$f.err
  python3 - $OUT/bench_$f.json $f <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
k = {x["kernel"]: x for x in d["kernels_survey_step"]}
ws = sum(v["ms_per_step"] for n, v in k.items() if "wgrad" in n)
print(f"slab_frac {sys.argv[2]}: {d['value']} img/s  {d['ms_per_step']} ms/step  kernel_ms {d['kernel_ms_per_step']}  colsum_batch {k.get('colsum_batch_kernel', {}).get('ms_per_step')}  wgrad kernels {ws:.3f}")
PY
done
