#!/bin/bash
# same-box A/B of the F(4x4,3x3) kernel: default library vs scripts/dbg/lib/libssdseg_w4_<tag>.so, `rounds` alternations; per run: forward /
# input-gradient ms and the traced cycles per item (loop, epilogue).  usage (gpurun, repo root): bash scripts/dbg/w4_ab_trace.sh <rounds> default tagA ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; shift
for ((i = 0; i < ROUNDS; ++i)); do
  for tag in "$@"; do
    if [ $tag = default ]; then unset SSDSEG_LIB; else export SSDSEG_LIB=$R/scripts/dbg/lib/libssdseg_w4_$tag.so; fi
    SSDSEG_W4_TRACE=1 timeout -k 10 120 python3 $R/scripts/conv3_decoder_time.py 3 > /tmp/w4ab.txt 2>&1
    echo "$tag: $(grep -E 'wino4_kernel' /tmp/w4ab.txt | awk '{print $3 $4, $7}' | tr '\n' ' ') | $(grep -E 'trace block   0: 40' /tmp/w4ab.txt | tail -1 | sed 's/.*items, //') | $(grep -E '^y:|^dx:' /tmp/w4ab.txt | awk '{print $NF}' | tr '\n' ' ')"
  done
done
