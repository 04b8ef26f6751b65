#!/bin/bash
# per-layer times (4 pages per launch) of the split-product conv variants: ASEP_SPLIT_ALDS = 0 | 1 | 2 | 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/alds
for D in "$@"; do
  ASEP_SPLIT_ALDS=$D ASEP_LAYER_PROFILE_PAGES=4 ASEP_F32_SPLIT=1 python3 scripts/gpu_layer_profile.py 4500 3000 f32 2 > gpurun_out/alds/l_$D.log 2>&1
  echo "mode=$D"; grep -E "total|down_3/convR_0|down_4/convR_0|down_2/convR_0|up_2/conv1|up_3/conv1|up_1/conv1" gpurun_out/alds/l_$D.log | cut -c1-140
done
