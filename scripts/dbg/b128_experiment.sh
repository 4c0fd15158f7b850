#!/bin/bash
# one GPU pass that pins the 16-byte-store anomaly of pw_wgrad_kernel<1,4,1,1>.
#   (1) scripts/micro/store_x4_hazard: the store + overwrite in inline assembly with 0/1/2 wait states, SGPR vs immediate soffset
#   (2) scripts/dbg/pww.py on four builds of the library (scripts/dbg/lib, made by hand from the same sources with -DPWW_STORE=n (0 for the two-store build)):
#       default (two 8-byte stores), 1 = one b128 store with SGPR soffset as the compiler schedules it, 2 = the same store
#       followed by s_nop 1 (inline asm), 3 = b128 store with the row offset in voffset (immediate soffset 0: the compiler's own
#       hazard recogniser then keeps two wait states)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b128
mkdir -p $OUT
cd $R/scripts/micro && hipcc --offload-arch=gfx950 -O3 -o store_x4_hazard store_x4_hazard.hip 2>/dev/null
timeout -k 10 300 ./store_x4_hazard > $OUT/hazard.txt 2>&1
cat $OUT/hazard.txt
cd $R
for shape in "70001 24 144" "614400 24 144" "2457600 16 96" "153600 32 192"; do
  for v in default 1 2 3; do
    if [ $v = default ]; then unset SSDSEG_LIB; else export SSDSEG_LIB=$R/scripts/dbg/lib/libssdseg_b128_$v.so; fi
    echo "== shape $shape  build $v" | tee -a $OUT/pww.txt
    timeout -k 10 300 python3 scripts/dbg/pww.py $shape 2>&1 | tee -a $OUT/pww.txt
  done
done
