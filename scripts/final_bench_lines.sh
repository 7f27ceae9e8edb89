#!/bin/bash
# GPU box: the bench lines of the round's evidence set (the second half of scripts/final_profiles.sh). usage: final_bench_lines.sh <tag>
tag=$1
set -e
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c2.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_c2_driver.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 --train-mode --no-cpu-baseline > gpurun_out/${tag}_bench_c2_train.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C3 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c3.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C5 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c5.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --episodes 128 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_bench_c2x128.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --episodes 32 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_bench_c2x32.json 2>/dev/null
echo done
