#!/bin/bash
# GPU box: the round's final evidence set under gpurun_out/ (copy into profiles/ afterwards):
# kernel stats + PMC passes (scripts/profile_bench.sh), bench lines, SQ and LDS counters.  usage: final_profiles.sh <tag>
tag=$1
bash scripts/profile_bench.sh ${tag}_c2 --config C2 || echo "profile_bench ${tag}_c2 stopped at pass $? (1 stats, 2 FETCH_SIZE, 3 WRITE_SIZE)"
bash scripts/profile_bench.sh ${tag}_c3 --config C3 || echo "profile_bench ${tag}_c3 stopped at pass $? (1 stats, 2 FETCH_SIZE, 3 WRITE_SIZE)"
bash scripts/profile_bench.sh ${tag}_c5 --config C5 || echo "profile_bench ${tag}_c5 stopped at pass $? (1 stats, 2 FETCH_SIZE, 3 WRITE_SIZE)"
bash scripts/profile_bench.sh ${tag}_c2x128 --config C2 --episodes 128 || echo "profile_bench ${tag}_c2x128 stopped at pass $? (1 stats, 2 FETCH_SIZE, 3 WRITE_SIZE)"
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c2.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_c2_driver.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 --train-mode --no-cpu-baseline > gpurun_out/${tag}_bench_c2_train.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C3 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c3.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C5 --steps 200 --warmup 20 > gpurun_out/${tag}_bench_c5.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --episodes 128 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_bench_c2x128.json 2>/dev/null
for c in c2 c3 c5; do
  C=$(echo $c | tr a-z A-Z)
  scripts/diag/sq_counters.sh ${tag}_$c --config $C
  scripts/diag/lds_counters.sh ${tag}_$c --config $C
done
echo done
