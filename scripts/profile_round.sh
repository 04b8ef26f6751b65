#!/bin/bash
# One measurement point of a build on the GPU box: rocprofv3 summaries + bench lines of the headline (fp32 engine default, f32s), of the bf16
# variant and of the plain fp32 kernels.  Started through scripts/profile_round_submit.sh, which refuses a dirty tree and passes HEAD's hash.
#   scripts/profile_round.sh <tag> <commit>      ->  gpurun_out/<tag>/, gpurun_out/<tag>_bf16/, gpurun_out/<tag>_f32/, gpurun_out/traffic_per_kernel*.json
# The profile script runs twice per dtype: the second traced bench line then carries the PMC traffic of the first (same build).
set -u
TAG=${1:-r3}
COMMIT=${2:-n/a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for T in "$TAG:" "${TAG}_bf16:--dtype bf16" "${TAG}_f32:--dtype f32"; do
    D=${T%%:*}; FL=${T#*:}
    mkdir -p gpurun_out/$D
    # the un-traced default line names the dominant kernel; the traced runs report the same one
    python3 bench.py $FL --no-secondary --no-cpu-baseline --steps 20 > gpurun_out/$D/bench_first.json 2> gpurun_out/$D/bench.err
    DOM=$(python3 -c "import json,sys; print(json.loads(open('gpurun_out/$D/bench_first.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
    for PASS in 1 2; do
        bash scripts/profile_bench.sh $D $FL --dominant "$DOM" > gpurun_out/${D}_profile.log 2>&1
        python3 scripts/make_traffic_json.py gpurun_out/$D $COMMIT > gpurun_out/$D/traffic.log 2>&1
    done
    python3 bench.py $FL > gpurun_out/$D/bench.json 2> gpurun_out/$D/bench.err
    DT=f32s; [ -n "$FL" ] && DT=${FL#--dtype }
    python3 scripts/gpu_layer_profile.py 4500 3000 $DT > gpurun_out/$D/layers_one_page.log 2>&1
done
cp profiles/traffic_per_kernel*.json gpurun_out/
bash scripts/pmc_instruction_mix.sh ${TAG}_mix > gpurun_out/${TAG}_mix.log 2>&1
bash scripts/pmc_instruction_mix.sh ${TAG}_mix_bf16 "--dtype=bf16" > gpurun_out/${TAG}_mix_bf16.log 2>&1
bash scripts/pmc_instruction_mix.sh ${TAG}_mix_f32 "--dtype=f32" > gpurun_out/${TAG}_mix_f32.log 2>&1
tail -2 gpurun_out/$TAG/bench.err
