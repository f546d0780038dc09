#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repository root); summaries land in gpurun_out/prof_r02/.
# usage: tools/profile_r02.sh <part>   with part = trace | pmc4096 | pmc1m | sq | calib
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
calib)
  # what rocprofv3 itself adds: a replayed graph of 512 dependent launches of a near-empty kernel, live period printed by the
  # script while it is being profiled, next to the dispatch duration the profiler reports for the same launches
  python3 tests/diag_profiler_calib.py 2>&1 | grep "us per launch" > $OUT/calib_unprofiled.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/cal -- python3 tests/diag_profiler_calib.py 2>&1 | grep "us per launch" > $OUT/calib_profiled.txt
  python3 profiles/summarize.py r02_calib_empty $RAW/cal --kernel k_pid_reset --out $OUT
  { echo "== un-profiled"; cat $OUT/calib_unprofiled.txt; echo "== the same script under rocprofv3 --kernel-trace --stats"; cat $OUT/calib_profiled.txt;
    echo "== rocprofv3's dispatch statistics of k_pid_reset in that run";
    python3 -c "import json; t=json.load(open('$OUT/r02_calib_empty_rocprof_summary.json'))['step_kernel_trace']; print('dispatches %d  average duration %.0f ns  median %.0f ns  median start-to-start %.0f ns' % (t['dispatches'], t['avg_ns'], t['median_ns'], t['median_start_to_start_ns']))"; } > $OUT/r02_profiler_calibration.txt
  ;;
esac
ls -la $OUT
