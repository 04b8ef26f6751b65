#!/bin/bash
# Per-layer times of the product library and of ablation builds (scripts/r5_abl_build.sh) on ONE box, base first and last.
#   scripts/r5_abl.sh <tag> <dtype> <grep pattern> lib1.so lib2.so ...      ->  gpurun_out/<tag>/
set -u
ulimit -c 0
TAG=$1; DT=$2; PAT=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
for L in base "$@" base; do
  N=$(basename $L .so)
  if [ $L = base ]; then unset ASEP_HIP_LIB; else export ASEP_HIP_LIB=$R/$L; fi
  ASEP_LAYER_PROFILE_PAGES=4 python3 scripts/gpu_layer_profile.py 4500 3000 $DT 5 > gpurun_out/$TAG/layers_$N.log 2>&1
  echo "== $N: $(head -2 gpurun_out/$TAG/layers_$N.log | grep total)"
  grep -E "$PAT" gpurun_out/$TAG/layers_$N.log
done | tee gpurun_out/$TAG/summary.txt
