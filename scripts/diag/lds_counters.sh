#!/bin/bash
# GPU box: LDS counters per kernel of the step (one --pmc pass), usage: lds_counters.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/lds_$tag -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-rotating --probe none --breakdown 0 "$@" > gpurun_out/lds_$tag.log 2>&1 || exit 1
ls gpurun_out/lds_$tag
