#!/bin/bash
# GPU box: first backward launch at K <= 32, tile-per-block kernel against the persistent kernel, by batch size.
export DENET_CLS_WIDE1_TILES=384
for n in 8 16 32 64 128; do
  for t in 0 100000000; do
    DENET_BWD_WIDE1_TILES=$t timeout -k 10 200 python3 bench.py --config C2 --episodes $n --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bw1_${n}_$t.json 2>gpurun_out/bw1_${n}_$t.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/bw1_${n}_$t.json').read().strip().splitlines()[-1])
k=d['kernels_us']
print($n, 'wide' if $t==0 else 'narrow', d['ms_per_step'], 'B1', k['bwd_dwn'], 'B2', k['bwd_chain_dfq'], 'B3', k['bwd_params_dfs'])"
  done
done
