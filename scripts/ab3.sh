#!/bin/bash
# like ab.sh at another config: scripts/ab3.sh <config> <variant>...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cfg=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset DENET_HIP_LIB; else export DENET_HIP_LIB=$PWD/scripts/diag/lib_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --config $cfg --steps 50 --no-cpu-baseline 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_us'])" || exit 1
done
