#!/bin/bash
# K3's large sort: round 4's count / scan / scatter passes (FHX_K3_SORT=legacy) against the one-sweep passes over all bits
# (FHX_OS_PASSES=8) and over the top 40 bits + repair (default), at three survivor fractions (lognormal rate noise s).
for s in ${SIGMAS:-1.0 2.0 3.0}; do
  echo "== --overdispersion $s"
  bash profiles/ab_env.sh "--overdispersion $s --steps 5 --warmup 1" "FHX_K3_SORT=legacy" "FHX_OS_PASSES=8" "FHX_OS_PASSES=6" "FHX_X=default" "FHX_OS_PASSES=4"
done
