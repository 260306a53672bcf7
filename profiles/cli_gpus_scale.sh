#!/bin/bash
# `fithic --gpus N` at C3 scale with the ranks sharing this box's one GPU (collectives over pipes): every rank reads the file and
# writes its stretches itself (default) against rows handed out and gathered by rank 0 (FHX_CLI_FUNNEL=1), next to --gpus 1.
# What it shows: the stages rank 0 used to do alone no longer grow with N.  (Wall times of N ranks on ONE GPU say nothing about N GPUs.)
D=${DIR:-/dev/shm/cli_gpus}
C=${CHROMS:-22}
python profiles/time_cli_scale.py --chroms $C --dir $D --tag g1 --md5
for N in 2 4; do
  DEV=$(python -c "print(','.join(['0']*$N))")
  echo "== --gpus $N, every rank reads and writes its own rows"
  FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag g${N} --md5
  echo "== --gpus $N, rows through rank 0 (FHX_CLI_FUNNEL=1)"
  FHX_CLI_FUNNEL=1 FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag f${N}
done
rm -rf $D
