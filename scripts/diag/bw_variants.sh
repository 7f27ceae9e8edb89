#!/bin/bash
# Diagnostic: first backward launch (B1) of C3 / C5 with pieces of the persistent K > 32 kernel removed
# (builds: hipcc -DDENET_DIAG_BW_<V> head_step.hip -> scripts/diag/build/libdenet_bw_<V>.so).
set -e
mkdir -p gpurun_out
for cfg in ${CFGS:-C3 C5}; do
  for v in ${VARIANTS:-BASE NOMFMA NODL NOSTORE ONLYDWN ONLYDFQ}; do
    if [ $v = BASE ]; then unset DENET_HIP_LIB; else export DENET_HIP_LIB=$PWD/scripts/diag/build/libdenet_bw_$v.so; fi
    timeout -k 10 120 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --probe bwd_dwn > gpurun_out/bw_${cfg}_$v.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/bw_${cfg}_$v.json')); print('$cfg $v', d['kernels_us']['bwd_dwn'], d['kernels_us']['bwd_chain_dfq'], d['ms_per_step'])"
  done
done
