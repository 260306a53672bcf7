#!/bin/bash
# k2h_heavy on small inputs: ONE launch with two rows per lane at eight waves per SIMD (round 4; FHX_K2H_SPLIT=0) against whole rounds of
# <4 rows, 4 waves> + the remainder as <1 row, 8 waves> (round 6) - the largest shard of an N-way sharding of C3 and C2
for N in 8 4 2; do echo "== largest shard of $N"; bash profiles/ab_env.sh "--shard-of $N --steps 40 --warmup 5" "FHX_K2H_SPLIT=0" "FHX_X=default"; done
echo "== C2"; bash profiles/ab_env.sh "--config C2 --steps 20 --warmup 3" "FHX_K2H_SPLIT=0" "FHX_X=default"
echo "== C3 (one launch either way: 1.48e8 rows)"; bash profiles/ab_env.sh "--steps 8 --warmup 2" "FHX_K2H_SPLIT=0" "FHX_X=default"
