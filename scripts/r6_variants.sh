#!/bin/bash
# one page of the elu / leaky ARU graphs on the bf16 engine: fused general blocks (default) against layer by layer (ASEP_FUSED8=0), per layer,
# each layer alone on the chip   ->  gpurun_out/r6_variants/<graph>_<form>.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r6_variants
python3 scripts/gpu_layer_profile.py 4500 3000 bf16 3 > gpurun_out/r6_variants/relu_ARU_bf16.log 2>&1
for A in elu leaky; do
  ASEP_LAYER_PROFILE_CFG="{\"activation_name\": \"$A\"}" python3 scripts/gpu_layer_profile.py 4500 3000 bf16 3 > gpurun_out/r6_variants/${A}_ARU_bf16_fused.log 2>&1
  ASEP_FUSED8=0 ASEP_LAYER_PROFILE_CFG="{\"activation_name\": \"$A\"}" python3 scripts/gpu_layer_profile.py 4500 3000 bf16 3 > gpurun_out/r6_variants/${A}_ARU_bf16_layer_by_layer.log 2>&1
done
grep -H "^total" gpurun_out/r6_variants/*.log | sed 's/.*r6_variants.//' | cut -c1-110
