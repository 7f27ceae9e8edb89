#!/bin/bash
# GPU box: SQ instruction counters of the loss kernels, full build and the five phase-truncated builds.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" _stop1 _stop2 _stop3 _stop4 _stop5; do
  timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/phase$v -- $R/scripts/diag/loss_bench$v ${PHASE_CFG:-8 21 41 321 5} > $R/gpurun_out/phase$v.log 2>&1 || exit 1
  grep -E "wce|pair|focal" $R/gpurun_out/phase$v.log
done
