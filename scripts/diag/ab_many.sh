#!/bin/bash
# Diagnostic: bench.py (C2 by default) with the shipped library and every alternative build named, two rounds,
# same box.  usage: ab_many.sh "<bench args>" <alt names...>   (alt = denet_amd/libdenet_alt_<name>.so)
args=$1; shift
for round in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then unset DENET_HIP_LIB; else export DENET_HIP_LIB=$PWD/denet_amd/libdenet_alt_$v.so; fi
    timeout -k 10 120 python bench.py $args --steps 200 --warmup 20 --no-cpu-baseline --no-dropin --no-rotating --probe none > gpurun_out/abm_$v.json 2>gpurun_out/abm_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/abm_$v.err; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/abm_$v.json')); k=d['kernels_us']; print('%-12s' % '$v', d['ms_per_step'], ' '.join('%s %.1f' % (n[:12], u) for n, u in k.items()), 'train', d['train_mode']['ms_per_step'], d['train_mode']['ms_per_step_eval_interleaved'])"
  done
done
