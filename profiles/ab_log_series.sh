# (record of a dropped experiment; result: profiles/history/r04_lg_log_series_ab.txt)  In incbet's tail the logarithm of the operand that is
# fl(1 - prior) from a 9-instruction series -(d + d^2 (1/2 + d/3 + d^2/4 + d^3/5 + d^4/6)), d = 1 - v exact, for d < 2^-10 (checked on
# the CPU against 60-digit logarithms: within 0.5 ulp, identical to glibc's log on 40 000 arguments), the library's log otherwise.
# "nolog" = a second library built with the series compiled out, chosen by FHX_LIB.
mkdir -p gpurun_out/r04
for V in series nolog series nolog; do
  if [ $V = nolog ]; then export FHX_LIB=$PWD/profiles/_ab_lib/libfithic_nolog.so; else unset FHX_LIB; fi
  FHX_BENCH_HASH=1 bash profiles/run_profile.sh r04/lg_$V --steps 8 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  echo "== $V"; grep "k2h_heavy\|k2_queue_by_count" gpurun_out/r04/lg_${V}_kernel_stats.txt
  python -c "import json; d=json.load(open('gpurun_out/r04/lg_${V}_bench.json')); print(d['ms_per_step'], d['kernels_ms']['k2_pvalue'], [d[k] for k in d if 'digest' in k or 'hash' in k])"
done
