#!/bin/bash
# `fithic` and `fithic --gpus N` on a C3-size contacts file that is ONE plain gzip stream (what `gzip` writes; 1.48e8 rows, 4.6 GB of
# text).  N ranks share this box's one GPU (collectives over pipes) and its host cores (cores / N inflating threads per rank).
# The ranks inflate the stream together - rank r its N-th of the compressed bytes (sharded._ingest_stream_parts) - against every rank
# inflating the whole file (FHX_CLI_STREAM_PARTS=0: the route until round 5).  Per rank: inflate seconds (FHX_TIMING lines of
# fhx_host_inflate_part / fhx_text_part_resolve, or "parallel gunzip"), peak host RSS.
D=${DIR:-/dev/shm/cli_plain}
C=${CHROMS:-22}
python profiles/time_cli_scale.py --chroms $C --dir $D --plain --tag g1 --md5 --check-rows 0
for N in ${NS:-2 4 8}; do
  DEV=$(python -c "print(','.join(['0']*$N))")
  echo "== --gpus $N, the ranks inflate the stream together"
  FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag s${N} --md5
done
for N in ${OLD_NS:-4}; do
  DEV=$(python -c "print(','.join(['0']*$N))")
  echo "== --gpus $N, every rank inflates the whole file (FHX_CLI_STREAM_PARTS=0)"
  FHX_CLI_STREAM_PARTS=0 FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag t${N} --md5
done
# the same with a FIXED number of host threads per rank (16): what a rank's inflate takes when its cores do not shrink with N
for N in ${FIXED_NS:-2 4 8}; do
  DEV=$(python -c "print(','.join(['0']*$N))")
  echo "== --gpus $N, 16 host threads per rank, the ranks inflate the stream together"
  FHX_CLI_THREADS_PER_RANK=16 FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag x${N} 2>&1 | grep -E "^fithic|^==|stage: contacts|fhx_host_inflate_part|fhx_text_part_resolve|peak host RSS"
done
echo "== --gpus 4, 16 host threads per rank, every rank inflates the whole file"
FHX_CLI_STREAM_PARTS=0 FHX_CLI_THREADS_PER_RANK=16 FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=0,0,0,0 python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus 4 --tag y4 2>&1 | grep -E "^fithic|^==|stage: contacts|parallel gunzip: 2|peak host RSS"
rm -rf $D
