#!/bin/bash
# A/B of an engine switch on the GPU box: bench lines (no kernel timing, no secondary) for every "ENV=VAL" given, per dtype.
#   scripts/r4_ab.sh <tag> "<pytest args or empty>" ENV=VAL [ENV=VAL ...]
set -u
TAG=${1:-r4ab}; shift
PYT=${1:-}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
if [ -n "$PYT" ]; then
    timeout 2400 python3 -m pytest $PYT -q -s > gpurun_out/$TAG/pytest.log 2>&1
    echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
    tail -4 gpurun_out/$TAG/pytest.log
fi
for KV in "$@"; do
    for DT in bf16 f32; do
        ST=60; [ $DT = bf16 ] && ST=160
        for REP in 1 2; do
            env $KV python3 bench.py --dtype $DT --no-secondary --no-cpu-baseline --kernel-timing none --steps $ST > gpurun_out/$TAG/bench_${DT}_${KV}_$REP.json 2>> gpurun_out/$TAG/bench.err
            python3 -c "import json;l=json.loads(open('gpurun_out/$TAG/bench_${DT}_${KV}_$REP.json').read().strip().splitlines()[-1]);print('$DT $KV rep $REP', l['value'])"
        done
    done
done
