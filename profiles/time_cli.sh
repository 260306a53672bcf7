#!/bin/bash
# end-to-end wall time of the drop-in CLI on the bundled hESC chr1 40 kb set (the reference: 30 s / 67 s)
D=tests/golden/data
python -c "import torch" 2>/dev/null   # page the image in so that the first timing is not dominated by cold imports
for args in "-p 1 -x All" "-p 2 -x intraOnly -t $D/hESC_chr1_w40000.bias.gz"; do
  rm -rf /tmp/cli_out
  t0=$(date +%s.%N)
  python -m fithic_amd -i $D/hESC_chr1_w40000.contacts.gz -f $D/hESC_chr1_w40000.frags.gz -o /tmp/cli_out -r 40000 -L 50000 -U 5000000 -b 50 $args > /tmp/cli.log 2>&1
  t1=$(date +%s.%N)
  echo "wall $(python -c "print(round($t1 - $t0, 2))") s   args: $args   rows: $(zcat /tmp/cli_out/FitHiC.spline_pass1.res40000.significances.txt.gz | wc -l)"
done
