#!/bin/bash
# Register / spill summary of the fused level-0 kernels: compiles aru_engine.hip to assembly (gfx950) and counts the
# instructions that matter (lane spills, scratch, packed FMAs, scalar loads).   usage: scripts/kernel_regs.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/../citlab-article-separation-new_amd/csrc"
OUT=${TMPDIR:-/tmp}/asep_regs; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c aru_engine.hip -o $OUT/aru_engine.o -save-temps=obj 2>/dev/null
S=$OUT/aru_engine-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$S" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S*res8\S*)(.*?)\.end_amdhsa_kernel", s, re.S):
    blk = m.group(2)
    g = lambda k: re.search(k + r" (\S+)", blk).group(1)
    name = m.group(1)
    body = s[s.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    cnt = {k: len(re.findall(k, body)) for k in ("v_writelane", "v_readlane", "scratch_", "v_pk_fma_f32", "v_mfma", "s_load_dwordx16", "ds_read_b128", "s_waitcnt", "s_barrier")}
    print(name[:48], "vgpr", g("next_free_vgpr"), "sgpr", g("next_free_sgpr"), "scratch", g("private_segment_fixed_size"), cnt)
PY
