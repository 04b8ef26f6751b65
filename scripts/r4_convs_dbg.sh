#!/bin/bash
# timing experiments on convs_kernel (ASEP_CONVS_DBG bits; results are wrong on purpose): per-layer times of a 4-page call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/convsdbg
for D in "$@"; do
  ASEP_CONVS_DBG=$D ASEP_LAYER_PROFILE_PAGES=4 ASEP_F32_SPLIT=1 python3 scripts/gpu_layer_profile.py 4500 3000 f32 2 > gpurun_out/convsdbg/l_$D.log 2>&1
  echo "dbg=$D"; grep -E "down_3/convR_0|down_4/convR_0|down_1/convR_0|down_2/convR_0|up_1/conv1|attPart/conv2" gpurun_out/convsdbg/l_$D.log | cut -c1-130
done
