#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun, from the repository root): un-profiled bench lines, cooperative vs
# single-wave step on the same box, in-kernel timeline, batch-size sweep.  Output: gpurun_out/evidence_r02/.
set -e
OUT=gpurun_out/evidence_r02; mkdir -p $OUT
python3 bench.py > $OUT/r02_bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/r02_bench_steps20.json 2> $OUT/bench_s20.err
{
for nr in 0 1; do
  QD_DIAG_NORESET=$nr QD_COOP_MAX_ENVS=0 timeout -k 10 120 python3 tests/diag_coop.py
  QD_DIAG_NORESET=$nr timeout -k 10 120 python3 tests/diag_coop.py
done
QD_DIAG_CONFIG=config5 QD_DIAG_ENVS=8192 QD_COOP_MAX_ENVS=0 timeout -k 10 120 python3 tests/diag_coop.py
QD_DIAG_CONFIG=config5 QD_DIAG_ENVS=8192 timeout -k 10 120 python3 tests/diag_coop.py
} 2>&1 | grep -v amdgpu.ids > $OUT/r02_coop_vs_singlewave.txt
if [ -f tests/_build/libqd_diag.so ]; then
  { QD_LIB=tests/_build/libqd_diag.so QD_DIAG_NORESET=1 timeout -k 10 120 python3 tests/diag_coop_stamps.py; QD_LIB=tests/_build/libqd_diag.so timeout -k 10 120 python3 tests/diag_coop_stamps.py; } 2>&1 | grep -v amdgpu.ids > $OUT/r02_coop_timeline.txt
fi
{ QD_DIAG_SIZES=4096,16384,65536,262144,1048576,4194304 timeout -k 10 300 python3 tests/diag_sweep.py; QD_DIAG_CONFIG=config5 QD_DIAG_SIZES=8192,1048576 timeout -k 10 200 python3 tests/diag_sweep.py; QD_DIAG_CONFIG=config2 QD_DIAG_SIZES=4096,1048576 timeout -k 10 200 python3 tests/diag_sweep.py; } 2>&1 | grep -v amdgpu.ids > $OUT/r02_env_count_sweep.txt
{ QD_DIAG_T=64,256,1024 timeout -k 10 200 python3 tests/diag_frag_len.py; QD_DIAG_CONFIG=config5 QD_DIAG_ENVS=8192 QD_DIAG_T=256,1024 timeout -k 10 200 python3 tests/diag_frag_len.py; } 2>&1 | grep -v amdgpu.ids > $OUT/r02_fragment_length.txt
if [ -x tests/_build/qd_latency ]; then
  { timeout -k 10 60 tests/_build/qd_latency 1024 192; timeout -k 10 60 tests/_build/qd_latency 1024 64; } > $OUT/r02_kernel_start_latency.txt 2>&1
fi
ls -la $OUT
