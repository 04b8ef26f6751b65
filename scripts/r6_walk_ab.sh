#!/bin/bash
# A/B of the level-0 UP block forms on ONE box (base = ASEP_BF_WALK=0: res8f_kernel tiles; new = the strip walker), interleaved base / new / base / new:
# per-layer medians of one page at 4 pages per launch and the bf16 step.     scripts/r6_walk_ab.sh <tag>   ->  gpurun_out/<tag>/
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
for REP in 1 2; do
  for V in base new; do
    [ -n "${WALK_BASE:-}" ] && BASEV=$WALK_BASE || BASEV=0
    if [ $V = base ]; then export ASEP_BF_WALK=$BASEV; else unset ASEP_BF_WALK; fi
    ASEP_LAYER_PROFILE_PAGES=4 python3 scripts/gpu_layer_profile.py 4500 3000 bf16 5 > gpurun_out/$TAG/layers_${V}_$REP.log 2>&1
    python3 bench.py --dtype bf16 --no-secondary --no-cpu-baseline --kernel-timing none --steps 60 > gpurun_out/$TAG/bench_${V}_$REP.json 2> gpurun_out/$TAG/bench_${V}_$REP.err
    python3 -c "import json;d=json.loads(open('gpurun_out/$TAG/bench_${V}_$REP.json').read().strip().splitlines()[-1]);print('bf16 $V $REP', d['value'], d['ms_per_step'])"
    grep -i "res8\|res16\|total\|page" gpurun_out/$TAG/layers_${V}_$REP.log | head -12
  done
done
unset ASEP_BF_WALK
