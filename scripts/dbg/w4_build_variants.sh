#!/bin/bash
# builds scripts/dbg/lib/libssdseg_w4_<tag>.so for each "<tag>:<extra hipcc flags>" argument (only gemm.hip is recompiled per variant)
# usage: bash scripts/dbg/w4_build_variants.sh "abl1:-DW4_ABL=1" "abl2:-DW4_ABL=2" ...
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/multi-task-learning-object-detection-semantic-segmentation_amd/csrc
mkdir -p $R/scripts/dbg/lib
make -C $C -j8 > /dev/null || exit 1
OTHERS=$(ls $C/*.o | grep -v '/gemm\.o$' | grep -v '/gemm_')
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -I$R/include -I$C $flags -c $C/gemm.hip -o /tmp/gemm_$tag.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dbg/lib/libssdseg_w4_$tag.so /tmp/gemm_$tag.o $OTHERS -ldl && echo "built $tag" ) &
done
wait
