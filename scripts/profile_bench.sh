#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + two separate PMC passes of the SAME bench
# command, outputs under gpurun_out/<tag>_{stats,fetch,write}.  Fold them afterwards with
# scripts/pmc_summary.py.  Usage: bash scripts/profile_bench.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
# (--settle 200: under the counter passes every dispatch is serialised; 2000 settle steps of C3 took minutes)
ARGS="--steps 100 --warmup 10 --settle 200 --no-cpu-baseline --no-dropin --no-rotating --probe none --breakdown 0 $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py $ARGS > gpurun_out/${tag}_stats.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py $ARGS > gpurun_out/${tag}_fetch.log 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py $ARGS > gpurun_out/${tag}_write.log 2>&1 || exit 3
# the per-dispatch trace is megabytes per run and gpurun copies back at most 64 MiB: keep the stats table only
find gpurun_out/${tag}_stats -name "*_kernel_trace.csv" -delete
ls gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
