#!/bin/bash
# AddressSanitizer + UBSan over everything that can run without a GPU: the C oracle and the device headers compiled for the
# host (tests/host_twin).  GPU sanitizers are not available on the pool.  Rebuilds the two test libraries instrumented, runs
# their test files, then restores the normal builds.
set -euo pipefail
cd "$(dirname "$0")/.."
SAN="-O1 -g -fPIC -fno-fast-math -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python -c "from oracle import oracle; oracle.build()"
mkdir -p tests/_build
tmp=$(mktemp -d)
cp -r oracle/_build "$tmp/oracle_build"
[ -f tests/_build/qd_host_twin.so ] && cp tests/_build/qd_host_twin.so "$tmp/twin.so"
restore() { rm -rf oracle/_build; mv "$tmp/oracle_build" oracle/_build; [ -f "$tmp/twin.so" ] && cp "$tmp/twin.so" tests/_build/qd_host_twin.so; rm -rf "$tmp"; }
trap restore EXIT
gcc $SAN -std=c11 -shared -o oracle/_build/libqd_oracle.so oracle/qd_oracle.c -lm
gcc $SAN -std=c11 -fopenmp -shared -o oracle/_build/libqd_oracle_omp.so oracle/qd_oracle.c -lm
g++ $SAN -std=c++17 -shared -I mujoco-drone_amd/csrc -o tests/_build/qd_host_twin.so tests/host_twin/qd_host_twin.cpp
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py tests/test_oracle_physics.py tests/test_host_twin.py -x -q
