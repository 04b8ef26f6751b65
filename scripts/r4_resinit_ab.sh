#!/bin/bash
# fp32: residual operand of the 16-channel layers as the accumulators' initial value (four blocks per CU) against the prefetch form (two)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/resinit
timeout 1200 python3 -m pytest tests/test_aru_gpu.py tests/test_aru_batch_gpu.py -m gpu -q -x 2>&1 | tail -3
for D in 0 1; do
  ASEP_RES_INIT=$D python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/resinit/l_$D.log 2>&1
  echo "ASEP_RES_INIT=$D"; grep -E "total|_1/convR_2" gpurun_out/resinit/l_$D.log | cut -c1-150
  for REP in 1 2; do
    ASEP_RES_INIT=$D python3 bench.py --no-secondary --no-cpu-baseline --kernel-timing none --steps 60 > gpurun_out/resinit/b_${D}_$REP.json 2>/dev/null
    python3 -c "import json;l=json.loads(open('gpurun_out/resinit/b_${D}_$REP.json').read().strip().splitlines()[-1]);print('bench f32 ASEP_RES_INIT=$D', l['value'])"
  done
done
