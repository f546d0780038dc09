// qd_rollout_fused32.hip -- k_rollout_fused_pipe (qd_rollout_fused.hip) once more with 32 envs per workgroup: two 16-env MFMA
// blocks that share every weight register (s_dense, qd_policy_static.h), the env roles on 32 lanes of their waves.
//
// Why: the closed loop keeps one workgroup per CU (512 threads at 256 registers), and what a pass costs is dominated by streaming
// the network's weights out of L2 (363 KB per pass for RMA_full, every workgroup its own copy: 25 TB/s across the chip at 4096
// envs) and by per-layer latencies, not by the MFMAs.  Past 256 workgroups of 16 envs the passes run in rounds; with 32 envs per
// workgroup 8192 envs are ONE round whose pass is ~1.4x as long, not two rounds (qd_rollout_policy picks this copy above 4096 envs).
//
// The whole unit is the other file compiled under a different tile constant.  Everything it defines lives in namespace qd, which is
// renamed here so that the two copies share no symbol; the host side calls this one through qd_fused32_launch (extern "C").
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define QD_POL_TILE 32
#define qd qd_t32
#include "qd_rollout_fused.hip"
