#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command, then two PMC passes
# (FETCH_SIZE, WRITE_SIZE) in their own runs, as /opt/skills/guides prescribe.  Output -> gpurun_out/<tag>/
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 2 pages per launch under the profiler (the default of 16 only lengthens the trace); bench.py scales the PMC bytes to its own batch
BENCH="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --pages-per-step 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-kernel-timing > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --no-kernel-timing > /dev/null 2> $OUT/pmc_write.log
# SQ / GRBM pass: MFMA pipe occupancy and wave stall buckets per kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH --no-kernel-timing > /dev/null 2> $OUT/pmc_sq.log
find $OUT -name "*.csv" | head -20
# keep only what fits the 64 MiB merge limit: drop the per-dispatch traces of the PMC passes after summarising
python3 $R/scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.json
python3 $R/scripts/summarize_pmc.py $OUT sq > $OUT/pmc_sq_summary.json
rm -f $OUT/pmc_fetch/*/*kernel_trace.csv $OUT/pmc_write/*/*kernel_trace.csv $OUT/pmc_sq/*/*kernel_trace.csv $OUT/pmc_sq/*/*counter_collection.csv
ls -la $OUT $OUT/*/* | head -40
