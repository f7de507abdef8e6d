#!/bin/bash
# rocprofv3 recipe for the step kernel: kernel-trace + stats, then PMC passes (never combined with tracing).
# usage: bash scripts/profile.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
# resident mode (the default): one worker launch per timed region, so warm-up and timed region get the SAME number of steps —
# the kernel-stats average over the two trs_worker_kernel dispatches is then the figure bench.py reports
# (--profile-mode: the time-based pre-warm runs by launches (trs_step_kernel, its own row), the worker leaves after the warm-up: three dispatches of 1000 steps each — warm-up, the
# wall-clock pass, the event-bracketed pass — so the stats table's average is a per-1000-steps figure like bench.py's avg_launch_us)
ARGS="--steps 1000 --warmup 1000 --no-cpu-baseline --no-also --profile-mode $@"
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
            "WRITE_SIZE" "FETCH_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"
done
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
