#!/bin/bash
# GPU box: the round's evidence set, FOLDED on the box into gpurun_out/profiles_out/ (gpurun copies back at most
# 64 MiB and the raw per-dispatch CSVs are ~4 MB per pass): kernel stats + PMC traffic (scripts/profile_bench.sh ->
# scripts/pmc_summary.py), bench lines, SQ and LDS counters (scripts/counter_fold.py).  Copy profiles_out/* into
# profiles/ afterwards.  usage: final_profiles.sh <tag>
tag=$1
out=gpurun_out/profiles_out
mkdir -p $out
# heartbeat: a run that writes nothing for three minutes is taken to be hung (counter passes are quiet for long)
( while sleep 45; do date +%T >> gpurun_out/heartbeat.txt; done ) &
hb=$!
prof() {   # prof <suffix> <config> <episodes> [bench args]
  local sfx=$1 cfg=$2 eps=$3; shift 3
  bash scripts/profile_bench.sh ${tag}_$sfx "$@" > gpurun_out/${tag}_${sfx}_profile.log 2>&1
  local rc=$?
  [ $rc -ne 0 ] && echo "profile_bench ${tag}_$sfx stopped at pass $rc (1 stats, 2 FETCH_SIZE, 3 WRITE_SIZE)"
  if [ $rc -eq 0 ]; then
    python3 scripts/pmc_summary.py ${tag}_$sfx gpurun_out/${tag}_${sfx}_stats gpurun_out/${tag}_${sfx}_fetch gpurun_out/${tag}_${sfx}_write --config $cfg --episodes $eps --out $out > /dev/null
  elif [ $rc -ne 1 ]; then
    python3 scripts/pmc_summary.py ${tag}_$sfx gpurun_out/${tag}_${sfx}_stats --config $cfg --episodes $eps --out $out > /dev/null
  fi
  rm -rf gpurun_out/${tag}_${sfx}_stats gpurun_out/${tag}_${sfx}_fetch gpurun_out/${tag}_${sfx}_write gpurun_out/${tag}_${sfx}_*.log
}
prof c2 C2 8 --config C2
prof c3 C3 20 --config C3
prof c5 C5 8 --config C5
prof c2x128 C2 128 --config C2 --episodes 128
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 > $out/${tag}_bench_c2.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 20 --warmup 5 > $out/${tag}_bench_c2_driver.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --steps 200 --warmup 20 --train-mode --no-cpu-baseline > $out/${tag}_bench_c2_train.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C3 --steps 200 --warmup 20 > $out/${tag}_bench_c3.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C5 --steps 200 --warmup 20 > $out/${tag}_bench_c5.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --episodes 32 --steps 200 --warmup 20 --no-cpu-baseline > $out/${tag}_bench_c2x32.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config C2 --episodes 128 --steps 200 --warmup 20 --no-cpu-baseline > $out/${tag}_bench_c2x128.json 2>/dev/null
: > $out/${tag}_sq_counters.txt
for c in c2 c3 c5; do
  C=$(echo $c | tr a-z A-Z)
  scripts/diag/sq_counters.sh ${tag}_$c --config $C > /dev/null 2>&1
  python3 scripts/counter_fold.py "SQ instruction counters, $C" gpurun_out/sq_${tag}_$c >> $out/${tag}_sq_counters.txt
  scripts/diag/lds_counters.sh ${tag}_$c --config $C > /dev/null 2>&1
  python3 scripts/counter_fold.py "LDS counters, $C" gpurun_out/lds_${tag}_$c >> $out/${tag}_sq_counters.txt
  rm -rf gpurun_out/sq_${tag}_$c gpurun_out/lds_${tag}_$c gpurun_out/sq_${tag}_$c.log gpurun_out/lds_${tag}_$c.log
done
kill $hb 2>/dev/null
rm -f gpurun_out/heartbeat.txt
du -sh gpurun_out
echo done
