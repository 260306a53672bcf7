#!/bin/bash
# `fithic --gpus N` at C3 scale, the two ways the ranks can split a contacts file they read themselves: parts of the FILE (default
# for a file of size-tagged members whose parts end on rows: each rank inflates and parses 1/N of it) against the split by chromosome
# (FHX_CLI_SPLIT=chromosome: every rank inflates and parses all of it and keeps its chromosomes' rows).  The ranks share this box's
# one GPU (collectives over pipes), so N parses of the whole file run one after another on it - the ingest stage shows exactly that.
D=${DIR:-/dev/shm/cli_gpus}
C=${CHROMS:-22}
python profiles/time_cli_scale.py --chroms $C --dir $D --tag g1 --md5
for N in 2 4 8; do
  DEV=$(python -c "print(','.join(['0']*$N))")
  echo "== --gpus $N, parts of the file"
  FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag s${N} --md5
  echo "== --gpus $N, split by chromosome (FHX_CLI_SPLIT=chromosome)"
  FHX_CLI_SPLIT=chromosome FHX_CLI_TRANSPORT=pipes FHX_CLI_DEVICES=$DEV python profiles/time_cli_scale.py --chroms $C --dir $D --reuse --gpus $N --tag c${N}
done
rm -rf $D
