#!/bin/bash
# The HOST half of librtfs_amd.so (scene flattening, the three tree builders, rt_scene_tune_rays, pixel maps, PPM) under
# AddressSanitizer + UndefinedBehaviorSanitizer: hipcc builds the host code instrumented (-Xarch_host; the device code is the
# ordinary gfx950 build -- GPU sanitizers are not available on this pool), and the CPU test suite runs against that library.
# usage: bash scripts/asan_host.sh      (here, no GPU needed; ~3 min to build)
set -eu
cd "$(dirname "$0")/.."
mkdir -p build_ab
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
( cd ray-tracing-fsharp_amd/csrc && hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
    -mllvm -simplifycfg-sink-common=false -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
    -shared-libsan -shared -o ../../build_ab/librtfs_asan.so rtfs_amd.hip -ldl )
RTFS_LIB=$PWD/build_ab/librtfs_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests -q -s -m "not gpu" -k "not in_tree_hip_build" > build_ab/asan_host.log 2>&1 || true
tail -1 build_ab/asan_host.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' build_ab/asan_host.log || true)"
