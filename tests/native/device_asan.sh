#!/bin/bash
# Device-side AddressSanitizer attempt: hipcc -fsanitize=address with an xnack+ code object.
#   1. a probe kernel that writes one element past its allocation - does the sanitizer report it on this box?
#   2. the whole library built the same way, driven by examples/ctypes_minimal.py (one small pass: K1, K2, K3) - a clean
#      run is the result asked for.
# Prints what happened at every step; run on the GPU box:  bash tests/native/device_asan.sh > gpurun_out/device_asan.txt 2>&1
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=/tmp/fhx_device_asan
mkdir -p $OUT
ASAN_RT=$(dirname $(hipcc -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null) 2>/dev/null)
[ -d "$ASAN_RT" ] || ASAN_RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1 | xargs dirname)
echo "== toolchain: $(hipcc --version | head -1); asan runtime dir: $ASAN_RT; device rtl: $(ls /opt/rocm/amdgcn/bitcode/asanrtl.bc 2>&1); instrumented ROCm libs: $(ls -d /opt/rocm/lib/asan 2>&1 | head -1)"
echo "== 1. probe (out-of-bounds store by one lane)"
hipcc -fsanitize=address -shared-libsan -g --offload-arch=gfx950:xnack+ -o $OUT/probe "$ROOT/tests/native/device_asan_probe.hip" 2>&1 | tail -3
echo "build rc=$?"
for xn in 1 0; do
  echo "-- HSA_XNACK=$xn"
  HSA_XNACK=$xn LD_LIBRARY_PATH=$ASAN_RT:$LD_LIBRARY_PATH timeout 120 $OUT/probe 2>&1 | head -25
  echo "run rc=${PIPESTATUS[0]}"
done
echo "== 2. the library, instrumented"
SRC="$ROOT/fithic_amd/csrc"
if [ -f "$ROOT/ab/libfithic_mi355x_asan.so" ]; then echo "(prebuilt in the build container with the command below: ab/libfithic_mi355x_asan.so)"; cp "$ROOT/ab/libfithic_mi355x_asan.so" $OUT/; else
( time hipcc -fsanitize=address -shared-libsan -g --offload-arch=gfx950:xnack+ -O1 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -pthread \
    -o $OUT/libfithic_mi355x_asan.so $SRC/fhx_device.hip $SRC/fhx_k1.hip $SRC/fhx_k2.hip $SRC/fhx_k3.hip $SRC/fhx_kr.hip $SRC/fhx_cni.hip $SRC/fhx_host.cpp $SRC/fhx_io.cpp $SRC/fhx_gunzip.cpp -lz -ldl ) 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | tail -8
fi
ls -la $OUT/libfithic_mi355x_asan.so 2>&1
echo "-- one small pass through ctypes (LD_PRELOAD of the sanitizer runtime: python itself is not instrumented)"
HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$ASAN_RT/libclang_rt.asan-x86_64.so timeout 600 python "$ROOT/examples/ctypes_minimal.py" $OUT/libfithic_mi355x_asan.so 2>&1 | tail -25
echo "run rc=${PIPESTATUS[0]}"
