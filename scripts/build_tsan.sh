#!/bin/bash
# ThreadSanitizer build of the host side (tests/test_gpu_sanitizers.py): the host code of libspartan_hip.so (completion polling, the ZK
# look-ahead thread, the few-term commitment engine, the process-wide table cache) and the whole host driver. Device code is untouched.
set -e
cd "$(dirname "$0")/.."
make -C spartan_amd/csrc variant NAME=tsan FLAGS="-Xarch_host -fsanitize=thread -Xarch_host -g" -j8 > /dev/null
CL=/opt/rocm/lib/llvm/bin/clang++
ccs=$(ls spartan_amd/host/*.cc | sort | tr '\n' ' ')
$CL -O1 -g -std=c++17 -fPIC -shared -fsanitize=thread -Wno-unknown-pragmas -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $ccs \
  -o spartan_amd/lib/libspartan_host_tsan.so -Lspartan_amd/lib -lspartan_hip_tsan -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
ls -la spartan_amd/lib/libspartan_hip_tsan.so spartan_amd/lib/libspartan_host_tsan.so
