#!/bin/bash
# A/B of two builds of libasep_hip.so on ONE box: per-layer medians of one page and the bench step, interleaved (base, new, base, new).
#   scripts/r5_ab.sh <tag> <base.so> [dtypes...]      ->  gpurun_out/<tag>/
set -u
TAG=$1; BASE=$2; shift 2
DTYPES=${@:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
for DT in $DTYPES; do
  for REP in 1 2; do
    for V in base new; do
      if [ $V = base ]; then export ASEP_HIP_LIB=$R/$BASE; else unset ASEP_HIP_LIB; fi
      ASEP_LAYER_PROFILE_PAGES=4 python3 scripts/gpu_layer_profile.py 4500 3000 $DT 5 > gpurun_out/$TAG/layers_${DT}_${V}_$REP.log 2>&1
      python3 bench.py --dtype $DT --no-secondary --no-cpu-baseline --kernel-timing none --steps 40 > gpurun_out/$TAG/bench_${DT}_${V}_$REP.json 2> gpurun_out/$TAG/bench_${DT}_${V}_$REP.err
      python3 -c "import json;d=json.loads(open('gpurun_out/$TAG/bench_${DT}_${V}_$REP.json').read().strip().splitlines()[-1]);print('$DT $V $REP', d['value'], d['ms_per_step'])"
    done
  done
done
unset ASEP_HIP_LIB
