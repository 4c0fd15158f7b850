#!/bin/bash
# same-box A/B of the decoder 3x3 conv kernels: default library vs scripts/dbg/lib/libssdseg_w4_<tag>.so, alternating, `reps` rounds
# usage (through gpurun, repo root): bash scripts/dbg/conv3_ab.sh <grep pattern> <rounds> default tagA tagB ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; ROUNDS=$2; shift 2
for ((i = 0; i < ROUNDS; ++i)); do
  for tag in "$@"; do
    if [ $tag = default ]; then unset SSDSEG_LIB; else export SSDSEG_LIB=$R/scripts/dbg/lib/libssdseg_w4_$tag.so; fi
    echo "$tag: $(timeout -k 10 120 python3 $R/scripts/conv3_decoder_time.py 5 2>&1 | grep -E "$PAT" | awk '{print $1, $2, $3, $4, $5, $6}' | tr '\n' ' ')"
  done
done
