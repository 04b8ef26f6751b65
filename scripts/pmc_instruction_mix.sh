#!/bin/bash
# Instruction mix / issue-activity counters per kernel (three rocprofv3 --pmc passes of 8 SQ counters each) -> gpurun_out/<tag>/
set -u
TAG=${1:-mix}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-gnn --no-kernel-timing --pages-per-step 2 ${2:-}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_a -- $BENCH > /dev/null 2> $OUT/pmc_a.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_b -- $BENCH > /dev/null 2> $OUT/pmc_b.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c -- $BENCH > /dev/null 2> $OUT/pmc_c.log
python3 - "$OUT" <<'PY' > $OUT/instruction_mix.json
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "pmc_*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        a = agg[row["Kernel_Name"]][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
res = {k: {c: v[1] / max(v[0], 1) for c, v in cs.items()} for k, cs in agg.items() if "asep" in k}
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
ls -la $OUT
