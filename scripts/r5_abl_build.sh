#!/bin/bash
# Ablation builds of libasep_hip.so (wrong results on purpose; see R8F_ABL in csrc/bf16_kernels.h): ab/libasep_hip_abl_<define>_<bits>.so
#   scripts/r5_abl_build.sh R8F_ABL 1 2 4 8 16        (only aru_engine.hip is recompiled; the other objects are the product's)
set -eu
DEF=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
C=$R/citlab-article-separation-new_amd/csrc
mkdir -p $R/build_abl $R/ab
(cd $C && make -s asep_common.o gnn_engine.o post_engine.o)
for B in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -D$DEF=$B -c $C/aru_engine.hip -o $R/build_abl/aru_engine_${DEF}_$B.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/build_abl/aru_engine_${DEF}_$B.o $C/asep_common.o $C/gnn_engine.o $C/post_engine.o -o $R/ab/libasep_hip_abl_${DEF}_$B.so && echo built $DEF=$B ) &
done
wait
