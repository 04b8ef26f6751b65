#!/bin/bash
# split-product fp32 path on the GPU box: the fp32 parity tests with ASEP_F32_SPLIT=1, per-layer timing, bench lines with and without.
#   scripts/r4_split.sh <tag> "<pytest args or empty>"
set -u
TAG=${1:-r4s}; shift
PYT=${1:-}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
if [ -n "$PYT" ]; then
    ASEP_F32_SPLIT=1 timeout 2400 python3 -m pytest $PYT -q -s -x > gpurun_out/$TAG/pytest.log 2>&1
    echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
    tail -15 gpurun_out/$TAG/pytest.log
fi
ASEP_F32_SPLIT=1 python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/$TAG/layers_split.log 2>&1
ASEP_LAYER_PROFILE_PAGES=4 ASEP_F32_SPLIT=1 python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/$TAG/layers_split4.log 2>&1
head -60 gpurun_out/$TAG/layers_split4.log | cut -c1-150
for KV in ASEP_F32_SPLIT=0 ASEP_F32_SPLIT=1; do
    env $KV python3 bench.py --no-secondary --no-cpu-baseline --kernel-timing none --steps 40 > gpurun_out/$TAG/bench_$KV.json 2>> gpurun_out/$TAG/bench.err
    python3 -c "import json;l=json.loads(open('gpurun_out/$TAG/bench_$KV.json').read().strip().splitlines()[-1]);print('$KV', l['value'])"
done
