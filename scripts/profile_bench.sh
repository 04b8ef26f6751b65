#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then two PMC passes (FETCH_SIZE, WRITE_SIZE) in
# their own runs, as /opt/skills/guides prescribe, then the SQ pass.  Output -> gpurun_out/<tag>/
#   scripts/profile_bench.sh <tag> [extra bench flags, e.g. --dtype bf16]
set -u
TAG=${1:-r3}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the default 16 pages per step: the same launch population as the un-traced default run (4 groups of 4 pages + the relation
# nets' grouped backbone), so per-kernel averages need no scaling;
# --kernel-timing in-situ: every launch of the traced process runs in the real schedule (side stream + relation nets), so
# rocprofv3's AverageNs and the line's HIP-event averages describe the same launches (scripts/roofline_from_profiles.py)
# 1 warm-up + 1 plain step, then 10 event-timed steps: 10 of the 12 steps rocprofv3 averages over are the ones the events bracket
# ASEP_LANES=1 (round 6): the product splits calls of >= 8 pages over two page lanes, and two launches of a kernel that share the chip take about twice as
# long each; a handle that records launch times runs on one lane anyway, but the warm-up and the plain step of this command would not -- with them on two
# lanes rocprofv3's AverageNs sat 5-20 % above the HIP-event averages of the same run (profiles/r6final first cut).  The whole traced process on one lane:
# both sources describe the same launches.  (The un-traced default lines, bench_first.json / bench.json, run the product's default.)
export ASEP_LANES=1
BENCH="python3 $R/bench.py --steps 1 --warmup 1 --event-steps 10 --no-cpu-baseline --no-secondary --kernel-timing in-situ $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.log
# SQ / GRBM pass: MFMA pipe occupancy and wave stall buckets per kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > /dev/null 2> $OUT/pmc_sq.log
# keep only what fits the 64 MiB merge limit: summarise, then drop the per-dispatch traces
python3 $R/scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.json
python3 $R/scripts/summarize_pmc.py $OUT sq > $OUT/pmc_sq_summary.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
python3 $R/scripts/roofline_from_profiles.py $OUT > $OUT/roofline_recomputed.json
ls -la $OUT
