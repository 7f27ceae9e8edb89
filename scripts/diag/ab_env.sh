#!/bin/bash
# GPU box: A/B of one environment knob on bench.py, alternating runs.  usage: ab_env.sh VAR A B [bench args...]
var=$1; a=$2; b=$3; shift 3
for rep in 1 2 3; do
  for v in $a $b; do
    env $var=$v timeout -k 10 200 python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --breakdown 0 --probe none "$@" > gpurun_out/ab_env.json 2>gpurun_out/ab_env.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_env.json').read().strip().splitlines()[-1])
print('$var=$v', d['ms_per_step'], d.get('train_mode',{}).get('ms_per_step'))"
  done
done
