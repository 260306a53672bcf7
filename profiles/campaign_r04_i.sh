#!/bin/bash
mkdir -p gpurun_out/r04
bash tests/native/device_asan.sh > gpurun_out/r04/i_device_asan.txt 2>&1
tail -40 gpurun_out/r04/i_device_asan.txt | cut -c1-220
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/i_c3_bench.json 2> gpurun_out/r04/i_c3_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r04/i_c3_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'], d.get('k3_stress'), d['parity_check']['ok'])"
