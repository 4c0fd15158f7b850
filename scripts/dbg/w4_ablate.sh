#!/bin/bash
# times the decoder 3x3 conv (scripts/conv3_decoder_time.py) under each variant library of scripts/dbg/lib/libssdseg_w4_<tag>.so
# usage (through gpurun, repo root): bash scripts/dbg/w4_ablate.sh default abl1 abl2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/w4_ablate.txt
: > $OUT
for tag in "$@"; do
  if [ $tag = default ]; then unset SSDSEG_LIB; else export SSDSEG_LIB=$R/scripts/dbg/lib/libssdseg_w4_$tag.so; fi
  echo "== $tag" >> $OUT
  SSDSEG_W4_TRACE=1 timeout -k 10 120 python3 $R/scripts/conv3_decoder_time.py 2 2>&1 | grep -E "trace block   0|wino4_kernel|^y:|^dx:" | sort | uniq | tail -12 >> $OUT || exit 1
done
cat $OUT
