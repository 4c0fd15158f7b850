#!/bin/bash
# per-layer A/B of the short-M column-tile rule of the tile GEMM (scripts/profile_ops.py full 32): A = one tile of <= 160 columns
# and the round-2 dispatch, B = default
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pwt
mkdir -p $O
export TOP=2000
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 $R/scripts/profile_ops.py full 32 > $O/$tag.txt 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }; echo "$tag done"; }
run A SSDSEG_PWT_SMALL=0
run B SSDSEG_NOTHING=1
python3 $R/scripts/ab_ops.py $O/A.txt $O/B.txt > $O/ab.txt
tail -1 $O/ab.txt
