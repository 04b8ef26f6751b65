#!/bin/bash
# round-4 measurement point on the GPU box: the GPU tests named on the command line, then traffic / timing of one dtype.
#   scripts/r4_check.sh <tag> "<pytest args>" [bench flags, e.g. --dtype bf16]
set -u
TAG=${1:-r4a}; shift
PYT=${1:-}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
if [ -n "$PYT" ]; then
    timeout 1500 python3 -m pytest $PYT -x -q -s > gpurun_out/$TAG/pytest.log 2>&1
    echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
    tail -5 gpurun_out/$TAG/pytest.log
fi
python3 bench.py "$@" --no-secondary --no-cpu-baseline --steps 20 > gpurun_out/$TAG/bench_first.json 2> gpurun_out/$TAG/bench.err
DOM=$(python3 -c "import json,sys; print(json.loads(open('gpurun_out/$TAG/bench_first.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
bash scripts/profile_bench.sh $TAG "$@" --dominant "$DOM" > gpurun_out/${TAG}_profile.log 2>&1
python3 scripts/make_traffic_json.py gpurun_out/$TAG r4 > gpurun_out/$TAG/traffic.log 2>&1
cp profiles/traffic_per_kernel*.json gpurun_out/$TAG/
python3 bench.py "$@" --no-secondary --no-cpu-baseline > gpurun_out/$TAG/bench.json 2>> gpurun_out/$TAG/bench.err
tail -3 gpurun_out/$TAG/bench.err
python3 - <<PY
import json
l=json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print(l["value"], l["roofline"])
PY
