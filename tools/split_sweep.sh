#!/bin/bash
# lab: sweep of the two-stream split point (VIT_AMD_SPLIT_PCT = share of the batch on the first stream), one box
# usage: tools/split_sweep.sh "50 56 62 68" [rounds] [bench args]
P=${1:-"50 43 47 53 57 62 38 33"}; R=${2:-2}; shift 2
for r in $(seq $R); do for p in $P; do
  echo -n "pct $p: "; VIT_AMD_SPLIT_PCT=$p timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernels "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.0f img/s' % (d['ms_per_step'], d['value']))" || exit 1
done; done
