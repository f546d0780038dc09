#!/bin/bash
# Round-4 profile collection on the GPU box (run through gpurun from the repository root); summaries land in gpurun_out/prof_r04/,
# tools/install_profiles_r04.py copies them into profiles/ and writes profiles/current.json (which bench.py reads, hash-checked).
# usage: tools/profile_r04.sh <part> [<part> ...]   with part = trace | pmc | sq | big | bigsq | c25 | policy
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04; RAW=/tmp/prof_raw; mkdir -p $OUT $RAW
B="--no-extras --no-cpu-baseline"
K=k_rollout_lat      # configs 3 and 5 at <= 16384 envs; larger batches: k_rollout_coop
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for part in "$@"; do
case "$part" in
trace)
  # the driver's command and the default command: kernel trace + stats (the persistent kernel: a few dozen dispatches of ~1.2 ms)
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/s20 -- python3 bench.py --steps 20 --warmup 5 $B > $OUT/bench_s20_profiled.json
  python3 profiles/summarize.py r04_s20_n4096 $RAW/s20 --kernel $K --out $OUT --grid 16384 --cut 20
  cp $(find $RAW/s20 -name "*_kernel_stats.csv" | head -1) $OUT/r04_s20_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/def -- python3 bench.py $B > $OUT/bench_default_profiled.json
  python3 profiles/summarize.py r04_default_n4096 $RAW/def --kernel $K --out $OUT --grid 16384 --cut 1024
  cp $(find $RAW/def -name "*_kernel_stats.csv" | head -1) $OUT/r04_default_kernel_stats.csv
  ;;
pmc)
  # HBM-side traffic of the persistent kernel on 1024-step fragments: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md)
  export QD_BENCH_RAMP_STEPS=2048
  A="--steps 2048 --warmup 1024"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/p0 -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/pf -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/pw -- python3 bench.py $A $B > /dev/null
  python3 profiles/summarize.py r04_pmc_n4096 $RAW/p0 --kernel $K --out $OUT --grid 16384 --cut 1024 --pmc fetch=$RAW/pf --pmc write=$RAW/pw
  ;;
sq)
  # issue-side counters of the same launches: wave-instructions issued, wave cycles, waits (what bounds the launch at 4096 envs)
  export QD_BENCH_RAMP_STEPS=2048
  A="--steps 2048 --warmup 1024"
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/q0 -- python3 bench.py $A $B > /dev/null
  rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $RAW/q1 -- python3 bench.py $A $B > /dev/null
  python3 profiles/summarize.py r04_sq_n4096 $RAW/q0 --kernel $K --out $OUT --grid 16384 --cut 1024 --pmc sq=$RAW/q1
  ;;
c25)
  # configs 2 and 5 at their BASELINE sizes through their fragment kernels: duration, HBM-side traffic, issue-side counters
  export QD_BENCH_RAMP_STEPS=2048
  A="--steps 2048 --warmup 1024"
  for cfg in "config5 8192 k_rollout_lat 32768" "config2 4096 k_rollout_pair 8192"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${1}t -- python3 bench.py --config $1 --envs $2 $A $B > $OUT/bench_${1}_profiled.json
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/${1}f -- python3 bench.py --config $1 --envs $2 $A $B > /dev/null
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/${1}w -- python3 bench.py --config $1 --envs $2 $A $B > /dev/null
    rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $RAW/${1}q -- python3 bench.py --config $1 --envs $2 $A $B > /dev/null
    python3 profiles/summarize.py r04_pmc_${1}_n$2 $RAW/${1}t --kernel "$3" --out $OUT --grid $4 --cut 1024 --pmc fetch=$RAW/${1}f --pmc write=$RAW/${1}w --pmc sq=$RAW/${1}q
  done
  ;;
big)
  # 2^20 envs, 64-step fragments: configs 3, 5 (k_rollout_coop, 4 x 2^20 threads) and 2 (k_rollout, 2^20 threads): duration, traffic,
  # and the issue-side counters that say what bounds these launches
  export QD_BENCH_RAMP_STEPS=64
  A="--envs 1048576 --fragment 64 --steps 128 --warmup 64"
  for cfg in "config3 k_rollout_coop 4194304" "config5 k_rollout_coop 4194304" "config2 k_rollout< 1048576"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/${1}bt -- python3 bench.py --config $1 $A $B > $OUT/bench_${1}_n1m_profiled.json
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/${1}bf -- python3 bench.py --config $1 $A $B > /dev/null
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/${1}bw -- python3 bench.py --config $1 $A $B > /dev/null
    rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $RAW/${1}bq -- python3 bench.py --config $1 $A $B > /dev/null
    python3 profiles/summarize.py r04_pmc_${1}_n1048576 $RAW/${1}bt --kernel "$2" --out $OUT --grid $3 --cut 64 --longest 64 --pmc fetch=$RAW/${1}bf --pmc write=$RAW/${1}bw --pmc sq=$RAW/${1}bq
  done
  ;;
policy)
  # the closed policy -> env loop (RMA_full actor, 4096 envs, 256-step fragments): k_rollout_fused_pipe, duration and matrix-pipe share
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/pol -- python3 tests/diag_fused_stamps.py 4096 > $OUT/r04_policy_loop_wall.txt
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $RAW/polq -- python3 tests/diag_fused_stamps.py 4096 > /dev/null
  python3 profiles/summarize.py r04_policy_loop_n4096 $RAW/pol --kernel k_rollout_fused_pipe --out $OUT --grid 131072 --cut 256 --longest 256 --tol 0.85 --pmc sq=$RAW/polq
  # train_LSTM.py's pair at its 8192 envs: CNNestimator on 23-value rows, 32 envs per workgroup (qd_rollout_fused32.hip)
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/polc -- python3 tests/diag_fused_stamps.py 8192 cnn > $OUT/r04_policy_loop_cnn_wall.txt
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $RAW/polcq -- python3 tests/diag_fused_stamps.py 8192 cnn > /dev/null
  python3 profiles/summarize.py r04_policy_loop_cnn_n8192 $RAW/polc --kernel k_rollout_fused_pipe --out $OUT --grid 131072 --cut 256 --longest 256 --tol 0.85 --pmc sq=$RAW/polcq
  ;;
esac
done
ls -la $OUT
