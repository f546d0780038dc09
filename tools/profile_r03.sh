#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun from the repository root); summaries land in gpurun_out/prof_r03/,
# tools/install_profiles_r03.py copies them into profiles/ and writes profiles/current.json (which bench.py reads, hash-checked).
# usage: tools/profile_r03.sh <part>   with part = trace | pmc | sq | big | c25
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r03; RAW=/tmp/prof_raw; mkdir -p $OUT $RAW
B="--no-extras --no-cpu-baseline"
K=k_rollout_coop
case "$1" in
trace)
  # the driver's command and the default command: kernel trace + stats (the persistent kernel: a few dozen dispatches of ~1.5 ms)
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/s20 -- python3 bench.py --steps 20 --warmup 5 $B > $OUT/bench_s20_profiled.json
  python3 profiles/summarize.py r03_s20_n4096 $RAW/s20 --kernel $K --out $OUT --grid 16384 --cut 20
  cp $(find $RAW/s20 -name "*_kernel_stats.csv" | head -1) $OUT/r03_s20_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/def -- python3 bench.py $B > $OUT/bench_default_profiled.json
  python3 profiles/summarize.py r03_default_n4096 $RAW/def --kernel $K --out $OUT --grid 16384 --cut 1024
  cp $(find $RAW/def -name "*_kernel_stats.csv" | head -1) $OUT/r03_default_kernel_stats.csv
  ;;
pmc)
  # HBM-side traffic of the persistent kernel on 1024-step fragments: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md)
  export QD_BENCH_RAMP_STEPS=2048
  A="--steps 2048 --warmup 1024"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/p0 -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/pf -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/pw -- python3 bench.py $A $B > /dev/null
  python3 profiles/summarize.py r03_pmc_n4096 $RAW/p0 --kernel $K --out $OUT --grid 16384 --cut 1024 --pmc fetch=$RAW/pf --pmc write=$RAW/pw
  ;;
sq)
  export QD_BENCH_RAMP_STEPS=2048
  A="--steps 2048 --warmup 1024"
  C="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/q0 -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $RAW/q1 -- python3 bench.py $A $B > /dev/null
  python3 profiles/summarize.py r03_sq_n4096 $RAW/q0 --kernel $K --out $OUT --grid 16384 --cut 1024 --pmc sq=$RAW/q1
  ;;
esac
ls -la $OUT
