#!/bin/bash
# lab: A/B of two builds of the library on one box (VIT_AMD_LIB selects the .so), alternating bench.py runs
# usage: tools/ab_lib.sh libA.so libB.so [rounds] [extra bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do for L in $A $B; do
  echo -n "$L: "; VIT_AMD_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernels "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.0f img/s' % (d['ms_per_step'], d['value']))" || exit 1
done; done
