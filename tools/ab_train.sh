#!/bin/bash
# lab: A/B of two builds of the library on the training step (bench.py --workload train), alternating on one box
# usage: tools/ab_train.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do for L in $A $B; do
VIT_AMD_LIB=$PWD/$L timeout -k 10 200 python bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '%.3f ms  %.0f img/s' % (d['ms_per_step'], d['value']))" || exit 1
done; done
