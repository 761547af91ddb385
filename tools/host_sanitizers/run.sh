#!/bin/bash
# Host pattern builders (sliced-ELL pattern, parallel Morton sort, window format, its symmetric form + checker) under
# AddressSanitizer / UBSan and ThreadSanitizer on a synthetic flat lattice mesh.  CPU only (GPU sanitizers are not available on
# the pool).  usage: tools/host_sanitizers/run.sh [cells per edge, default 40: 384k tet10, uses the parallel radix sort]
set -e
N=${1:-40}
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../fcvm-workbench_amd/csrc
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I$SRC $HERE/pattern_driver.cpp $SRC/fcvm_pattern.cpp -o $T/asan
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -I$SRC $HERE/pattern_driver.cpp $SRC/fcvm_pattern.cpp -o $T/tsan
echo "== ASan + UBSan"; $T/asan $N
echo "== TSan"; $T/tsan $N
rm -rf $T
