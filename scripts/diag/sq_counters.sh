#!/bin/bash
# GPU box: SQ instruction counters per kernel of the step (one --pmc pass, <= 8 SQ counters), usage: sq_counters.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/sq_$tag -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-rotating --probe none --breakdown 0 "$@" > gpurun_out/sq_$tag.log 2>&1 || exit 1
ls gpurun_out/sq_$tag
