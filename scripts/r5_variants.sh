#!/bin/bash
# one page of the ARU graph variants (ARU_v1.py:43,70-75,228-233) against the default graph, every arithmetic (round 5: the bf16 engine
# serves them too), per layer, each layer alone on the chip   ->  gpurun_out/r5_variants/<graph>_<dtype>.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5_variants
for DT in f32s f32 bf16; do
  python3 scripts/gpu_layer_profile.py 4500 3000 $DT 3 > gpurun_out/r5_variants/relu_ARU_$DT.log 2>&1
  ASEP_LAYER_PROFILE_CFG='{"activation_name": "elu"}' python3 scripts/gpu_layer_profile.py 4500 3000 $DT 3 > gpurun_out/r5_variants/elu_ARU_$DT.log 2>&1
  ASEP_LAYER_PROFILE_CFG='{"activation_name": "leaky"}' python3 scripts/gpu_layer_profile.py 4500 3000 $DT 3 > gpurun_out/r5_variants/leaky_ARU_$DT.log 2>&1
  ASEP_LAYER_PROFILE_CFG='{"graph": "U"}' python3 scripts/gpu_layer_profile.py 4500 3000 $DT 3 > gpurun_out/r5_variants/relu_U_$DT.log 2>&1
  ASEP_LAYER_PROFILE_CFG='{"graph": "RU"}' python3 scripts/gpu_layer_profile.py 4500 3000 $DT 3 > gpurun_out/r5_variants/relu_RU_$DT.log 2>&1
done
grep -H "^total" gpurun_out/r5_variants/*.log | sed 's/.*r5_variants.//' | cut -c1-110
