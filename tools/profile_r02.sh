#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repository root); summaries land in gpurun_out/prof_r02/.
# usage: tools/profile_r02.sh <part>   with part = trace | pmc4096 | pmc1m | sq
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02; RAW=/tmp/prof_raw; mkdir -p $OUT $RAW
B="--no-extras --no-cpu-baseline"
case "$1" in
trace)
  # the driver's command and the default command, kernel trace + stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/s20 -- python3 bench.py --steps 20 --warmup 5 $B > $OUT/bench_s20_profiled.json
  python3 profiles/summarize.py r02_s20_n4096 $RAW/s20 --kernel k_step_coop --out $OUT
  cp $(find $RAW/s20 -name "*_kernel_stats.csv" | head -1) $OUT/r02_s20_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/def -- python3 bench.py $B > $OUT/bench_default_profiled.json
  python3 profiles/summarize.py r02_default_n4096 $RAW/def --kernel k_step_coop --out $OUT
  cp $(find $RAW/def -name "*_kernel_stats.csv" | head -1) $OUT/r02_default_kernel_stats.csv
  ;;
pmc4096)
  export QD_GRAPH_MIN_STEPS=2000000000 QD_BENCH_RAMP_STEPS=1024
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/p0 -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/pf -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/pw -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  python3 profiles/summarize.py r02_pmc_n4096 $RAW/p0 --kernel k_step_coop --out $OUT --pmc fetch=$RAW/pf --pmc write=$RAW/pw
  ;;
pmc1m)
  export QD_GRAPH_MIN_STEPS=2000000000 QD_BENCH_RAMP_STEPS=64
  A="--envs 1048576 --fragment 8 --steps 128 --warmup 16"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/m0 -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/mf -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/mw -- python3 bench.py $A $B > /dev/null
  python3 profiles/summarize.py r02_pmc_n1m $RAW/m0 --kernel "k_step_wide<" --out $OUT --pmc fetch=$RAW/mf --pmc write=$RAW/mw
  ;;
sq)
  export QD_GRAPH_MIN_STEPS=2000000000 QD_BENCH_RAMP_STEPS=1024
  C="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/q0 -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $RAW/q1 -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  python3 profiles/summarize.py r02_sq_coop_n4096 $RAW/q0 --kernel k_step_coop --out $OUT --pmc sq=$RAW/q1
  export QD_COOP_MAX_ENVS=0
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/q2 -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $RAW/q3 -- python3 bench.py --steps 1024 --warmup 64 $B > /dev/null
  python3 profiles/summarize.py r02_sq_singlewave_n4096 $RAW/q2 --kernel "k_step<" --out $OUT --pmc sq=$RAW/q3
  ;;
esac
ls -la $OUT
