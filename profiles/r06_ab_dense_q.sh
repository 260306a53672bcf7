#!/bin/bash
# K3's q through the dense array + k3_fill_q (FHX_K3_DENSE=1) against the scattered stores into the q column (FHX_K3_DENSE=0):
# the headline (77 k survivors), the three k3_stress workloads (12 / 39 / 55 % survivors), a 1/8 shard and C2.
B="--steps 8 --warmup 2"
echo "== C3 headline";            bash profiles/ab_env.sh "$B" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
echo "== lognormal s = 1.0";      bash profiles/ab_env.sh "--overdispersion 1.0 $B" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
echo "== hotspots 0.2:4.5";       bash profiles/ab_env.sh "--hotspots 0.2:4.5 $B" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
echo "== hotspots 0.25:3.9";      bash profiles/ab_env.sh "--hotspots 0.25:3.9 $B" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
echo "== 1/8 shard";              bash profiles/ab_env.sh "--shard-of 8 --steps 30 --warmup 5" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
echo "== C2";                     bash profiles/ab_env.sh "--config C2 --steps 20 --warmup 3" "FHX_K3_DENSE=0" "FHX_K3_DENSE=1"
