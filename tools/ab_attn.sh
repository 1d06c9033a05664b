#!/bin/bash
# lab: A/B of two builds of the library on tools/time_attn.py, alternating on one box
# usage: tools/ab_attn.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do for L in $A $B; do
  echo "== $L"; VIT_AMD_LIB=$PWD/$L timeout -k 10 120 python tools/time_attn.py 2>/dev/null || exit 1
done; done
