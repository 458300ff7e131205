#!/bin/bash
# ASan + UBSan on the CPU side (GPU sanitizers are not available on the pool):
#  1. the C oracle, through its KAT / golden / gloo tests
#  2. the host code of libloudscan_hip.so (engine, scan.h layer, WAV reader, ebur128 shim;
#     device code uninstrumented), through the CPU tests that load it
set -e
cd "$(dirname "$0")/.."
make -C oracle -s -B CFLAGS="-O1 -g -std=gnu99 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle_kat.py tests/test_golden_cpu.py tests/test_album_gloo.py -x -q
make -C oracle -s -B
( cd loudgain_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined \
    -fno-gpu-sanitize -shared -o /tmp/libloudscan_hip_asan.so lgd_kernels.hip lgd_epilogue.hip lgd_engine.cpp scan_api.cpp ebur128_shim.cpp )
ASAN_OPTIONS=detect_leaks=0 LOUDSCAN_LIB=/tmp/libloudscan_hip_asan.so \
  LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
  python -m pytest tests/test_wav_reader_cpu.py tests/test_cabi_symbols.py tests/test_batch_host.py -x -q
