#!/bin/bash
# Diagnostic: bench.py with the shipped library and with an alternative build (DENET_HIP_LIB), same box.
# usage: ab_lib.sh <alt .so> <configs...>
set -e
alt=$1; shift
for cfg in "$@"; do EXTRA=""; if [ "$cfg" = C2x128 ]; then cfg=C2; EXTRA="--episodes 128"; fi
  for v in base alt base alt; do
    if [ $v = base ]; then unset DENET_HIP_LIB; else export DENET_HIP_LIB=$PWD/$alt; fi
    timeout -k 10 120 python bench.py --config $cfg $EXTRA --steps 200 --warmup 20 --no-cpu-baseline --probe none > gpurun_out/ab_${cfg}_$v.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_${cfg}_$v.json')); k=d['kernels_us']; print('$cfg $v', d['ms_per_step'], 'F1', k['masked_pool_fwd'], 'B3', k['bwd_params_dfs'], 'train', d['train_mode']['ms_per_step'])"
  done
done
