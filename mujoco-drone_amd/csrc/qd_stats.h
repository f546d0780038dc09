// qd_stats.h -- what the reference does with a finished train batch, on the device.
//
// custom_logging.py:9-31 (MyCallbacks.on_learn_on_batch) takes train_batch['obs'] / ['actions'] to the host and logs
// np.min / np.max / np.mean / np.var per column; training.py:16-22 reads RLlib's episode returns and lengths
// (episode_reward_mean, episode_len_mean, sum(episode_reward) / sum(episode_lengths)).  Here both run over the rollout
// fragments where they lie in HBM: k_column_stats streams a [rows, cols] float32 matrix once (HBM-bound: 4 bytes per
// element, nothing written but a few KB of partials), k_episode_stats walks reward / truncated [T, N] with one env per lane.
//
// Determinism: every reduction has a fixed order (lane slots -> waves -> workgroups), no atomics; sums are carried in
// float64, so the result is at least as accurate as numpy's float32 pairwise sums the reference logs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qd {

constexpr int STAT_THREADS = 256, STAT_UNROLL = 4, STAT_MAX_COLS = 64, STAT_GROUPS = 1024, STAT_FINAL_THREADS = 1024, STAT_FINAL_COLS = 8;
constexpr int EPI_UNROLL = 16;
typedef float stat_f4 __attribute__((ext_vector_type(4)));

// NaN-propagating min / max like numpy's: once NaN, always NaN
__device__ __forceinline__ float stat_min(float a, float v) { return (v < a || v != v) ? v : a; }
__device__ __forceinline__ float stat_max(float a, float v) { return (v > a || v != v) ? v : a; }
__device__ __forceinline__ double stat_min(double a, double v) { return (v < a || v != v) ? v : a; }
__device__ __forceinline__ double stat_max(double a, double v) { return (v > a || v != v) ? v : a; }

__host__ __device__ constexpr int stat_gcd4(int cols) { return cols % 4 == 0 ? 4 : cols % 2 == 0 ? 2 : 1; }
// threads of a workgroup that carry data: the largest P <= 256 with 4 P = 0 (mod cols)
__host__ __device__ constexpr int stat_period(int cols) { return STAT_THREADS / (cols / stat_gcd4(cols)) * (cols / stat_gcd4(cols)); }

// The matrix is read as a flat array of 16-byte units (`head` = 0..3 leading floats up to the first 16-byte boundary and up to
// 3 trailing floats are left to the final kernel).  Thread t < P of a workgroup reads units t, t + P, t + 2P, ... of the
// workgroup's span: since 4 P is a multiple of cols, its four lanes of a unit ALWAYS hold columns (4 t + k + head) mod cols,
// k = 0..3 -- four fixed columns per thread, accumulators in registers, every load a full coalesced dwordx4 (P >= 192 of the
// 256 threads active for any cols <= 64; 253 for the 22- and 23-wide observations).
// part[(group * cols + col) * 4 + {0,1,2,3}] = sum, sum of squares, min, max of what this workgroup saw.
__global__ __launch_bounds__(STAT_THREADS) void k_column_stats(const stat_f4* __restrict__ x4, long long units, int cols, int head,
                                                               double* __restrict__ part) {
  __shared__ double l_sum[4][STAT_THREADS], l_sq[4][STAT_THREADS], m_sum[STAT_THREADS], m_sq[STAT_THREADS];
  __shared__ float l_min[4][STAT_THREADS], l_max[4][STAT_THREADS], m_min[STAT_THREADS], m_max[STAT_THREADS];
  const int P = stat_period(cols), tid = threadIdx.x;
  const long long span = (long long)P * STAT_UNROLL;  // units per workgroup per iteration, contiguous
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  float mn[4], mx[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { mn[k] = __builtin_inff(); mx[k] = -__builtin_inff(); }
  if (tid < P)
    for (long long u0 = (long long)blockIdx.x * span + tid; u0 < units; u0 += (long long)gridDim.x * span) {
      stat_f4 v[STAT_UNROLL];
#pragma unroll
      for (int u = 0; u < STAT_UNROLL; u++) {
        const long long i = u0 + (long long)u * P;
        v[u] = __builtin_nontemporal_load(x4 + (i < units ? i : u0));  // out of range: re-read a valid unit, not accumulated
      }
#pragma unroll
      for (int u = 0; u < STAT_UNROLL; u++)
        if (u0 + (long long)u * P < units) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float f = v[u][k];
            const double d = (double)f;
            s[k] += d;
            q[k] = fma(d, d, q[k]);
            mn[k] = stat_min(mn[k], f);
            mx[k] = stat_max(mx[k], f);
          }
        }
    }
#pragma unroll
  for (int k = 0; k < 4; k++) { l_sum[k][tid] = s[k]; l_sq[k][tid] = q[k]; l_min[k][tid] = mn[k]; l_max[k][tid] = mx[k]; }
  __syncthreads();
  // fold the 4 P (thread, lane) entries per column in three fixed-order stages: J1 = 256 / cols slices of the entries of a
  // column (entries (t, k) with (4 t + k + head) mod cols == c, in increasing 4 t + k) -> at most 8 slices -> one
  const int J1 = STAT_THREADS / cols, J2 = J1 < 8 ? J1 : 8;
  {
    const int j = tid / cols, c = tid - j * cols;
    double ts = 0.0, tq = 0.0;
    float tmn = __builtin_inff(), tmx = -__builtin_inff();
    if (j < J1) {
      int e = c - head % cols;
      if (e < 0) e += cols;
      for (e += j * cols; e < 4 * P; e += J1 * cols) {
        const int t = e >> 2, k = e & 3;
        ts += l_sum[k][t]; tq += l_sq[k][t]; tmn = stat_min(tmn, l_min[k][t]); tmx = stat_max(tmx, l_max[k][t]);
      }
    }
    m_sum[tid] = ts; m_sq[tid] = tq; m_min[tid] = tmn; m_max[tid] = tmx;
  }
  __syncthreads();
  if (tid < J2 * cols) {  // the entry arrays are free now: slot 0 takes the second stage
    const int j = tid / cols, c = tid - j * cols;
    double ts = 0.0, tq = 0.0;
    float tmn = __builtin_inff(), tmx = -__builtin_inff();
    for (int i = j; i < J1; i += J2) {
      ts += m_sum[i * cols + c]; tq += m_sq[i * cols + c]; tmn = stat_min(tmn, m_min[i * cols + c]); tmx = stat_max(tmx, m_max[i * cols + c]);
    }
    l_sum[j >> 2][(j & 3) * cols + c] = ts; l_sq[j >> 2][(j & 3) * cols + c] = tq;
    l_min[j >> 2][(j & 3) * cols + c] = tmn; l_max[j >> 2][(j & 3) * cols + c] = tmx;
  }
  __syncthreads();
  if (tid < cols) {
    double ts = 0.0, tq = 0.0;
    float tmn = __builtin_inff(), tmx = -__builtin_inff();
    for (int j = 0; j < J2; j++) {
      const int a = j >> 2, b = (j & 3) * cols + tid;
      ts += l_sum[a][b]; tq += l_sq[a][b]; tmn = stat_min(tmn, l_min[a][b]); tmx = stat_max(tmx, l_max[a][b]);
    }
    double* o = part + ((size_t)blockIdx.x * cols + tid) * 4;
    o[0] = ts; o[1] = tq; o[2] = (double)tmn; o[3] = (double)tmx;
  }
}

// Workgroup b folds columns [8 b, 8 b + 8) over all groups (slices of groups in parallel, then the slices in order), adds the
// up to 3 + 3 floats outside the 16-byte units, and writes
// out[0..cols) = min, [cols..2cols) = max, [2cols..3cols) = mean, [3cols..4cols) = population variance (np.var, ddof 0)
__global__ __launch_bounds__(STAT_FINAL_THREADS) void k_column_stats_final(const double* __restrict__ part, int groups, int cols,
                                                                           long long rows, const float* __restrict__ x, int head, long long units,
                                                                           double* __restrict__ out) {
  constexpr int SL = STAT_FINAL_THREADS / STAT_FINAL_COLS;
  __shared__ double l_sum[STAT_FINAL_THREADS], l_sq[STAT_FINAL_THREADS], l_min[STAT_FINAL_THREADS], l_max[STAT_FINAL_THREADS];
  const int k = threadIdx.x / STAT_FINAL_COLS, cc = threadIdx.x - k * STAT_FINAL_COLS, c = blockIdx.x * STAT_FINAL_COLS + cc;
  double s = 0.0, q = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
  if (c < cols) {
    constexpr int U = 4;
    for (int g0 = k; g0 < groups; g0 += U * SL) {
      double2 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int g = g0 + u * SL;
        const double2* p = reinterpret_cast<const double2*>(part + ((size_t)(g < groups ? g : k) * cols + c) * 4);
        a[u] = p[0]; b[u] = p[1];
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (g0 + u * SL < groups) {
          s += a[u].x; q += a[u].y; mn = stat_min(mn, b[u].x); mx = stat_max(mx, b[u].y);
        }
    }
  }
  l_sum[threadIdx.x] = s; l_sq[threadIdx.x] = q; l_min[threadIdx.x] = mn; l_max[threadIdx.x] = mx;
  __syncthreads();
  if (k < 8 && c < cols) {  // 128 slices -> 8 -> 1, each in slice order
    s = 0.0; q = 0.0; mn = __builtin_inf(); mx = -__builtin_inf();
    for (int j = k; j < SL; j += 8) {
      const int i = j * STAT_FINAL_COLS + cc;
      s += l_sum[i]; q += l_sq[i]; mn = stat_min(mn, l_min[i]); mx = stat_max(mx, l_max[i]);
    }
  }
  __syncthreads();
  if (k < 8) { l_sum[threadIdx.x] = s; l_sq[threadIdx.x] = q; l_min[threadIdx.x] = mn; l_max[threadIdx.x] = mx; }
  __syncthreads();
  if (k == 0 && c < cols) {
    for (int j = 1; j < 8; j++) {
      const int i = j * STAT_FINAL_COLS + cc;
      s += l_sum[i]; q += l_sq[i]; mn = stat_min(mn, l_min[i]); mx = stat_max(mx, l_max[i]);
    }
    // the floats before the first / after the last 16-byte unit
    const long long total = rows * cols, head_n = head < total ? head : total, tail0 = head_n + 4 * units;
    for (long long e = 0; e < head_n; e++)
      if ((int)(e % cols) == c) { const double d = (double)x[e]; s += d; q = fma(d, d, q); mn = stat_min(mn, d); mx = stat_max(mx, d); }
    for (long long e = tail0; e < total; e++)
      if ((int)(e % cols) == c) { const double d = (double)x[e]; s += d; q = fma(d, d, q); mn = stat_min(mn, d); mx = stat_max(mx, d); }
    const double n = (double)rows, mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;  // rounding only: the exact value is non-negative
    out[c] = mn; out[cols + c] = mx; out[2 * cols + c] = mean; out[3 * cols + c] = var;
  }
}

// Episode bookkeeping over one fragment.  An episode ends at the step whose `truncated` flag is set (its reward counts,
// BaseDroneEnv.py:276-284).  One env per lane would be a 1024-step latency chain on 64 waves, so the fragment is cut into S
// time segments walked in parallel (k_episode_segments: one lane per (segment, env); reads of a step are coalesced across
// the envs) and stitched per env afterwards (k_episode_stitch).  A segment reports
//   head = (return, length) from its first step up to and including its first episode end (the whole segment if none ends),
//   tail = (return, length) after its last episode end, and the statistics of the episodes that start AND end inside it;
// the stitch adds the running episode carried in (carry[2n], carry[2n+1], from the previous fragment) to the heads.
// part[group * 8 + ...] = episodes, sum return, sum length, sum return^2, min return, max return, min length, max length
constexpr int EPI_FIELDS = 8, EPI_MAX_SEGMENTS = 64, EPI_MIN_SEGMENT = 8;
struct EpiAcc {
  double cnt = 0, sr = 0, sl = 0, sr2 = 0, mnr = __builtin_inf(), mxr = -__builtin_inf(), mnl = __builtin_inf(), mxl = -__builtin_inf();
  __device__ __forceinline__ void add(double ret, double len) {
    cnt += 1.0; sr += ret; sl += len; sr2 = fma(ret, ret, sr2);
    mnr = ret < mnr ? ret : mnr; mxr = ret > mxr ? ret : mxr;
    mnl = len < mnl ? len : mnl; mxl = len > mxl ? len : mxl;
  }
};
__device__ __forceinline__ double epi_merge(int f, double a, double b) {
  return f < 4 ? a + b : (f == 4 || f == 6) ? (b < a ? b : a) : (b > a ? b : a);
}
// the workgroup's accumulators, plus `n_extra` partial rows extra[j * extra_stride + field] (nullptr: none), folded in a fixed
// order: 32 slices per field in parallel, then the slices in order
constexpr int EPI_SLICES = STAT_THREADS / EPI_FIELDS;
__device__ __forceinline__ void epi_fold(const EpiAcc& a, double (*l)[STAT_THREADS], double (*l2)[EPI_SLICES], const double* __restrict__ extra,
                                         int n_extra, size_t extra_stride, double* __restrict__ part_row) {
  const int tid = threadIdx.x;
  l[0][tid] = a.cnt; l[1][tid] = a.sr; l[2][tid] = a.sl; l[3][tid] = a.sr2;
  l[4][tid] = a.mnr; l[5][tid] = a.mxr; l[6][tid] = a.mnl; l[7][tid] = a.mxl;
  __syncthreads();
  const int f = tid & (EPI_FIELDS - 1), k = tid / EPI_FIELDS;
  double v = l[f][k];
  for (int i = k + EPI_SLICES; i < STAT_THREADS; i += EPI_SLICES) v = epi_merge(f, v, l[f][i]);
  if (extra)
    for (int j = k; j < n_extra; j += EPI_SLICES) v = epi_merge(f, v, extra[(size_t)j * extra_stride + f]);
  l2[f][k] = v;
  __syncthreads();
  if (tid < EPI_FIELDS) {
    double w = l2[tid][0];
    for (int i = 1; i < EPI_SLICES; i++) w = epi_merge(tid, w, l2[tid][i]);
    part_row[tid] = w;
  }
}

// grid (ceil(N / 256), S); segment j covers steps [j L, min(T, (j + 1) L)).  seg_ret[(j N + n) * 2 + {0, 1}] = head / tail
// return, seg_len[...] = head length / tail length, tail length -1 when no episode ends inside the segment.
__global__ __launch_bounds__(STAT_THREADS) void k_episode_segments(const float* __restrict__ reward, const uint8_t* __restrict__ truncated,
                                                                   int T, int N, int L, double* __restrict__ seg_ret,
                                                                   int* __restrict__ seg_len, double* __restrict__ part) {
  __shared__ double l[EPI_FIELDS][STAT_THREADS], l2[EPI_FIELDS][EPI_SLICES];
  const int n = blockIdx.x * STAT_THREADS + threadIdx.x, j = blockIdx.y;
  const int t_begin = j * L, t_end = (t_begin + L < T) ? t_begin + L : T;
  EpiAcc acc;
  if (n < N) {
    double ret = 0.0, head_ret = 0.0;
    int len = 0, head_len = 0;
    bool ended = false;
    for (int t0 = t_begin; t0 < t_end; t0 += EPI_UNROLL) {
      float r[EPI_UNROLL];
      uint8_t e[EPI_UNROLL];
#pragma unroll
      for (int u = 0; u < EPI_UNROLL; u++) {
        const bool ok = t0 + u < t_end;
        r[u] = ok ? reward[(size_t)(t0 + u) * N + n] : 0.f;
        e[u] = ok ? truncated[(size_t)(t0 + u) * N + n] : (uint8_t)0;
      }
#pragma unroll
      for (int u = 0; u < EPI_UNROLL; u++) {
        if (t0 + u < t_end) {
          ret += (double)r[u];
          len += 1;
          if (e[u]) {
            if (!ended) { head_ret = ret; head_len = len; ended = true; }
            else acc.add(ret, (double)len);
            ret = 0.0; len = 0;
          }
        }
      }
    }
    const size_t at = ((size_t)j * N + n) * 2;
    seg_ret[at] = ended ? head_ret : ret; seg_ret[at + 1] = ret;
    seg_len[at] = ended ? head_len : len; seg_len[at + 1] = ended ? len : -1;
  }
  epi_fold(acc, l, l2, nullptr, 0, 0, part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EPI_FIELDS);
}

__global__ __launch_bounds__(STAT_THREADS) void k_episode_stitch(const double* __restrict__ seg_ret, const int* __restrict__ seg_len, int S,
                                                                 int N, double* __restrict__ carry, const double* __restrict__ seg_part,
                                                                 double* __restrict__ part) {
  __shared__ double l[EPI_FIELDS][STAT_THREADS], l2[EPI_FIELDS][EPI_SLICES];
  const int n = blockIdx.x * STAT_THREADS + threadIdx.x;
  EpiAcc acc;
  if (n < N) {
    double ret = carry[2 * n], len = carry[2 * n + 1];
    constexpr int U = 8;
    for (int j0 = 0; j0 < S; j0 += U) {
      double2 rr[U];
      int2 ll[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const size_t at = ((size_t)(j0 + u < S ? j0 + u : j0) * N + n) * 2;
        rr[u] = *reinterpret_cast<const double2*>(seg_ret + at);
        ll[u] = *reinterpret_cast<const int2*>(seg_len + at);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (j0 + u < S) {
          ret += rr[u].x; len += (double)ll[u].x;
          if (ll[u].y >= 0) {
            acc.add(ret, len);
            ret = rr[u].y; len = (double)ll[u].y;
          }
        }
    }
    carry[2 * n] = ret; carry[2 * n + 1] = len;
  }
  // this env group's S segment partials are folded in here, so the final kernel sees one row per env group
  epi_fold(acc, l, l2, seg_part + (size_t)blockIdx.x * EPI_FIELDS, S, (size_t)gridDim.x * EPI_FIELDS, part + (size_t)blockIdx.x * EPI_FIELDS);
}

// out[0..8) as `part`, then out[8] = mean return (episode_reward_mean), out[9] = mean length (episode_len_mean),
// out[10] = sum return / sum length (training.py:18), out[11] = population std of the returns; NaN where no episode ended
__global__ __launch_bounds__(STAT_THREADS) void k_episode_stats_final(const double* __restrict__ part, int groups, double* __restrict__ out) {
  __shared__ double l[STAT_THREADS];
  {  // 32 slices of the partials per field in parallel, then the slices in order
    const int f = threadIdx.x & (EPI_FIELDS - 1), k = threadIdx.x / EPI_FIELDS, SL = STAT_THREADS / EPI_FIELDS;
    double a = f < 4 ? 0.0 : (f == 4 || f == 6) ? __builtin_inf() : -__builtin_inf();
    for (int g = k; g < groups; g += SL) {
      const double b = part[(size_t)g * EPI_FIELDS + f];
      a = f < 4 ? a + b : (f == 4 || f == 6) ? (b < a ? b : a) : (b > a ? b : a);
    }
    l[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < EPI_FIELDS) {
      for (int j = 1; j < SL; j++) {
        const double b = l[j * EPI_FIELDS + f];
        a = f < 4 ? a + b : (f == 4 || f == 6) ? (b < a ? b : a) : (b > a ? b : a);
      }
      out[f] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double cnt = out[0], nan = __builtin_nan("");
    const double mean = cnt > 0 ? out[1] / cnt : nan;
    out[8] = mean;
    out[9] = cnt > 0 ? out[2] / cnt : nan;
    out[10] = out[2] > 0 ? out[1] / out[2] : nan;
    double var = cnt > 0 ? out[3] / cnt - mean * mean : nan;
    if (var < 0.0) var = 0.0;
    out[11] = cnt > 0 ? sqrt(var) : nan;
    if (!(cnt > 0)) out[4] = out[5] = out[6] = out[7] = nan;
  }
}

}  // namespace qd
