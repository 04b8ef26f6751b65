#!/bin/bash
# one page of the ARU graph variants (ARU_v1.py:43,70-75,228-233) against the default graph, per layer
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/variants
python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/relu_ARU.log 2>&1
ASEP_LAYER_PROFILE_CFG='{"activation_name": "elu"}' python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/elu_ARU.log 2>&1
ASEP_LAYER_PROFILE_CFG='{"activation_name": "leaky"}' python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/leaky_ARU.log 2>&1
ASEP_LAYER_PROFILE_CFG='{"graph": "U"}' python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/relu_U.log 2>&1
ASEP_LAYER_PROFILE_CFG='{"graph": "RU"}' python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/relu_RU.log 2>&1
head -1 gpurun_out/variants/*.log
# round 4, late: the level-0 blocks of the elu / leaky RESIDUAL graphs on res8v_*_kernel<activation> (default) against layer by layer
for A in elu leaky; do
  ASEP_FUSED8_VAR=0 ASEP_LAYER_PROFILE_CFG="{\"activation_name\": \"$A\"}" python3 scripts/gpu_layer_profile.py 4500 3000 f32 3 > gpurun_out/variants/${A}_ARU_level0_layer_by_layer.log 2>&1
done
head -1 gpurun_out/variants/*level0*.log
grep -h "res8v" gpurun_out/variants/elu_ARU.log gpurun_out/variants/leaky_ARU.log | cut -c1-100
