#!/bin/bash
# A/B kernel variants: scripts/ab.sh <variant>...  (libs built as scripts/diag/lib_<variant>.so; "base" = shipped lib)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for v in "$@"; do
  if [ "$v" = base ]; then unset DENET_HIP_LIB; else export DENET_HIP_LIB=$PWD/scripts/diag/lib_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_us'])" || exit 1
done
