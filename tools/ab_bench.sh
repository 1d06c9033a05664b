#!/bin/bash
# A/B of an environment switch on one box: alternating bench.py runs (forward, ViT-B/16 batch 256).
# usage: tools/ab_bench.sh VAR [rounds] [value0 value1]
VAR=$1; R=${2:-3}; V0=${3:-0}; V1=${4:-1}
for i in $(seq $R); do for v in $V0 $V1; do
  echo -n "$VAR=$v: "; env $VAR=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernels 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.0f img/s' % (d['ms_per_step'], d['value']))" || exit 1
done; done
