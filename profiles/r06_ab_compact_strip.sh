#!/bin/bash
# k3_compact: one tile of 16 384 rows and one atomic per workgroup (FHX_K3_TILES_PER_WG=1) against up to four tiles per workgroup with
# the survivors of sparse tiles held in an LDS strip and ONE atomic behind the last tile (default)
B="--steps 10 --warmup 3"
echo "== C3 headline";        bash profiles/ab_env.sh "$B" "FHX_K3_TILES_PER_WG=1" "FHX_X=default" "FHX_K3_TILES_PER_WG=8"
echo "== lognormal s = 1.0";  bash profiles/ab_env.sh "--overdispersion 1.0 $B" "FHX_K3_TILES_PER_WG=1" "FHX_X=default"
echo "== hotspots 0.25:3.9";  bash profiles/ab_env.sh "--hotspots 0.25:3.9 $B" "FHX_K3_TILES_PER_WG=1" "FHX_X=default"
echo "== 1/8 shard";          bash profiles/ab_env.sh "--shard-of 8 --steps 40 --warmup 5" "FHX_K3_TILES_PER_WG=1" "FHX_X=default" "FHX_K3_TILES_PER_WG=2"
echo "== C5 (wide totals)";   bash profiles/ab_env.sh "--config C5 --steps 3 --warmup 1" "FHX_K3_TILES_PER_WG=1" "FHX_X=default"
