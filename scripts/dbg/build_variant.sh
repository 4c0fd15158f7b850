#!/bin/bash
# builds scripts/dbg/lib/libssdseg_w4_<tag>.so with ONE translation unit recompiled under extra flags
# usage: bash scripts/dbg/build_variant.sh <file.hip> "<tag>:<extra hipcc flags>" ...      (select at run time with SSDSEG_LIB=...)
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/multi-task-learning-object-detection-semantic-segmentation_amd/csrc
F=$1; shift
B=${F%.hip}
mkdir -p $R/scripts/dbg/lib
make -C $C -j8 > /dev/null || exit 1
OTHERS=$(ls $C/*.o | grep -v "/$B\.o$")
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -I$R/include -I$C $flags -c $C/$F -o /tmp/${B}_$tag.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dbg/lib/libssdseg_w4_$tag.so /tmp/${B}_$tag.o $OTHERS -ldl && echo "built $tag" ) &
done
wait
