#!/bin/bash
# timing experiments on res8s_kernel (ASEP_R8S_DBG bits; results are wrong on purpose): per-layer time of the two level-0 blocks
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r8sdbg
for D in "$@"; do
  ASEP_R8S_DBG=$D ASEP_LAYER_PROFILE_PAGES=4 ASEP_F32_SPLIT=1 python3 scripts/gpu_layer_profile.py 4500 3000 f32 2 > gpurun_out/r8sdbg/l_$D.log 2>&1
  echo "dbg=$D"; grep res8s gpurun_out/r8sdbg/l_$D.log | cut -c1-70
done
