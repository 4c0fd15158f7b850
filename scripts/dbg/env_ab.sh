#!/bin/bash
# same-box A/B of the full train step under one environment setting: bash scripts/dbg/env_ab.sh <rounds> VAR=a VAR=b ...  ("-" = nothing set)
ROUNDS=$1; shift
for ((i = 0; i < ROUNDS; ++i)); do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then
      r=$(python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1)
    else
      r=$(env "$kv" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1)
    fi
    echo "$kv: $(echo "$r" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done
