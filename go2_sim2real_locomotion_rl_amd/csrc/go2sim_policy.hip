// go2sim_policy.hip -- policy inference next to the env step (include/go2sim_policy.h).  gfx950 only.
//
// One kernel evaluates a whole MLP: a workgroup (4 wavefronts) owns RT x 16 rows of the batch and keeps their activations in LDS across all
// layers; a layer is a [16 x K] x [K x N] product per row tile on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32: a k-ordered fma chain),
// the four wavefronts take the 16-column output tiles round-robin, bias and ELU are applied on the accumulator and the result goes to the other
// LDS buffer.  Weights stream from L2 (0.8 MB actor / 0.9 MB critic); every weight fragment a wavefront loads is used for RT row tiles (RT = 2: 32 rows
// per workgroup, 132 KB of LDS, one workgroup per CU, 2 x 128 workgroups at 4096 rows; the accumulation chain of every output element is the one of
// RT = 1).  Counters at 4096 rows (tools/policy_pmc.sh): 65 us per launch, the matrix pipe busy 38 % of the time (1.69 M matrix instructions = the
// 3.3 GFLOP), 23 % in the 6.5 M other vector instructions (mostly the deterministic exp of the ELU epilogue, which a wavefront runs between its
// matrix loops), the rest waiting (LDS / L2 round trips and the barriers between layers).  Measured without effect on that split: the row tile
// (RT = 1 at two workgroups per CU / RT = 2), K steps fetched 1 / 2 / 4 / 8 ahead (MLP_KU), a hand-issued double buffer.
// At 4096 rows: 3.3 GFLOP for actor + critic.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/go2sim.h"
#include "../../include/go2sim_detmath.h"
#include "../../include/go2sim_policy.h"

namespace {

#ifndef MLP_KU
#define MLP_KU 1
#endif
#ifndef MLP_RT
#define MLP_RT 2
#endif
// weights and biases are reached through pointers stored in the MlpDev record: without the address space the compiler issues FLAT loads, whose
// counters do not retire in order, and then waits for every outstanding load before each use
constexpr int MAXL = GO2SIM_MLP_MAX_LAYERS, MAXW = GO2SIM_MLP_MAX_WIDTH, RT = MLP_RT, TM = 16 * RT, NWAVE = 4, LDW = MAXW + 4;
constexpr uint32_t RNG_POLICY_NOISE = 11;   // purposes 1..10 belong to the environment (csrc/go2sim.hip)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gcfp;
typedef const __attribute__((address_space(1))) f32x4* gcf4p;

struct MlpDev {
  int n_layers;
  int din[MAXL], dout[MAXL], kpad[MAXL], npad[MAXL];
  const float* W[MAXL];   // [npad][kpad], zero padded
  const float* b[MAXL];   // [npad], zero padded
};

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) { fprintf(stderr, "go2sim_policy: %s failed: %s\n", #x, hipGetErrorString(e_)); return GO2SIM_E_HIP; } \
  } while (0)

__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : dm_exp(v) - 1.0f; }   // nn.ELU(alpha=1)

// One layer for the RT x 16 rows of the workgroup.  A wavefront works on TG output tiles (16 columns each) at a time: RT x TG independent accumulators
// share one weight fragment per (tile, K step) over the row tiles and one A fragment per (row tile, K step) over the output tiles.
template <int TG>
__device__ __forceinline__ void mlp_layer(const MlpDev& M, int l, const float (*in)[LDW], float (*out)[LDW], float* __restrict__ y, int row0, int B, int wave, int lane) {
  const int K = M.kpad[l], N = M.npad[l], dout = M.dout[l], ntiles = N / 16;
  const gcfp W = (gcfp)M.W[l];
  const gcfp bias = (gcfp)M.b[l];
  const bool last = l == M.n_layers - 1;
  const int arow = lane & 15, kq = lane >> 4;
  const float* ap = &in[arow][4 * kq];
  for (int g0 = wave * TG; g0 < ntiles; g0 += NWAVE * TG) {
    f32x4 acc[RT][TG];
    gcfp wrow[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      const int nt = (g0 + t < ntiles) ? g0 + t : ntiles - 1;          // a short last group recomputes the last tile (not stored)
      wrow[t] = W + (size_t)(nt * 16 + arow) * K + 4 * kq;
    }
    // K steps in groups of KU: the weight fragments of a whole group are requested first, then the matrix instructions of its steps are issued in
    // order, each step waiting only for its own fragments (the counter retires in order), so the L2 latency of the weight stream is paid once per
    // group instead of once per step -- one wavefront per SIMD has nothing else to hide it.  (A hand-issued double buffer across the loop back edge
    // was tried: the register copies the compiler places on that edge read the buffer before its loads have landed.)
    constexpr int KU = MLP_KU;
    for (int j0 = 0; j0 < K; j0 += 16 * KU) {
      f32x4 wv[KU][TG];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int j = (j0 + 16 * u < K) ? j0 + 16 * u : K - 16;         // (a short last group re-reads the last step: no branch around the loads)
#pragma unroll
        for (int t = 0; t < TG; ++t) wv[u][t] = *(gcf4p)(wrow[t] + j);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (j0 + 16 * u < K) {
          float4 av[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) av[rt] = *(const float4*)(ap + (size_t)rt * 16 * LDW + j0 + 16 * u);
#pragma unroll
          for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, wv[u][t][0], acc[rt][t], 0, 0, 0);   // k = j + 4 q + 0, q = 0..3
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, wv[u][t][1], acc[rt][t], 0, 0, 0);   // k = j + 4 q + 1
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, wv[u][t][2], acc[rt][t], 0, 0, 0);
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, wv[u][t][3], acc[rt][t], 0, 0, 0);
            }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (g0 + t >= ntiles) break;
      const int n = (g0 + t) * 16 + arow;                              // accumulator element i of this lane: row 4 * (lane >> 4) + i, column lane & 15
      const float bv = bias[n];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 16 * rt + 4 * kq + i;
          const float v = acc[rt][t][i] + bv;
          if (last) {
            const int gr = row0 + r;
            if (gr < B && n < dout) y[(size_t)gr * dout + n] = v;
          } else {
            out[r][n] = elu1(v);
          }
        }
    }
  }
}

// blockIdx.y selects the network: actor and critic of one policy step share a launch (2 x 128 workgroups of 32 rows at 4096 rows, one per CU)
__global__ __launch_bounds__(64 * NWAVE) void k_mlp_forward(MlpDev M0, const float* __restrict__ x0, float* __restrict__ y0,
                                                            MlpDev M1, const float* __restrict__ x1, float* __restrict__ y1, int B) {
  __shared__ alignas(16) float act[2][TM][LDW];
  const MlpDev& M = blockIdx.y == 0 ? M0 : M1;
  const float* __restrict__ x = blockIdx.y == 0 ? x0 : x1;
  float* __restrict__ y = blockIdx.y == 0 ? y0 : y1;
  const int row0 = blockIdx.x * TM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  {
    const int K0 = M.kpad[0], d0 = M.din[0];
    for (int idx = tid; idx < TM * K0; idx += 64 * NWAVE) {
      int r = idx / K0, k = idx - r * K0, gr = row0 + r;
      act[0][r][k] = (gr < B && k < d0) ? x[(size_t)gr * d0 + k] : 0.0f;
    }
  }
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < M.n_layers; ++l) {
    const int tiles_per_wave = (M.npad[l] / 16 + NWAVE - 1) / NWAVE;
    if (tiles_per_wave >= 4) mlp_layer<4>(M, l, act[cur], act[cur ^ 1], y, row0, B, wave, lane);
    else if (tiles_per_wave >= 2) mlp_layer<2>(M, l, act[cur], act[cur ^ 1], y, row0, B, wave, lane);
    else mlp_layer<1>(M, l, act[cur], act[cur ^ 1], y, row0, B, wave, lane);
    __syncthreads();
    cur ^= 1;
  }
}

// Normal(mean, std).sample() + log_prob, one lane per row (A <= 16 actions)
__global__ __launch_bounds__(64) void k_policy_sample(const float* __restrict__ mean, const float* __restrict__ std_, int B, int A, uint64_t seed, uint32_t step,
                                                      int deterministic, float* __restrict__ actions, float* __restrict__ log_prob) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float lp = 0.0f;
  for (int blk = 0; 4 * blk < A; ++blk) {
    float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!deterministic) {
      dm_u4 r = dm_philox((uint32_t)b, step, RNG_POLICY_NOISE, (uint32_t)blk, (uint32_t)seed, (uint32_t)(seed >> 32));
      dm_normal2(r.v[0], r.v[1], &n[0], &n[1]); dm_normal2(r.v[2], r.v[3], &n[2], &n[3]);
    }
    for (int k = 0; k < 4; ++k) {
      const int a = 4 * blk + k;
      if (a >= A) break;
      const float mu = mean[(size_t)b * A + a], sd = std_[a];
      const float act = mu + sd * n[k];
      actions[(size_t)b * A + a] = act;
      const float d = act - mu;
      lp = lp + ((-(d * d) / (2.0f * (sd * sd)) - dm_log(sd)) - 0.91893853320467274178f);
    }
  }
  if (log_prob) log_prob[b] = lp;
}

// ---- rollout storage (rsl_rl RolloutStorage.compute_returns) ----
__global__ __launch_bounds__(64) void k_rollout_add(float* __restrict__ rew, float* __restrict__ val, uint8_t* __restrict__ don, int B, const float* __restrict__ r,
                                                    const uint8_t* __restrict__ d, const float* __restrict__ v, const float* __restrict__ to, float gamma) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float rr = r[b];
  const float vv = v[b];
  if (to) rr = rr + gamma * (vv * to[b]);                               // PPO.process_env_step: bootstrap on time-outs
  rew[b] = rr; val[b] = vv; don[b] = d[b];
}
constexpr int GAE_WG = 256;
// one lane per env walks the rollout backwards; per-workgroup moments of the advantages in a fixed order
__global__ __launch_bounds__(GAE_WG) void k_gae(const float* __restrict__ rew, const float* __restrict__ val, const uint8_t* __restrict__ don, const float* __restrict__ last_values,
                                                int T, int B, float gamma, float lam, float* __restrict__ ret, float* __restrict__ adv, double* __restrict__ partial) {
  __shared__ double s_sum[GAE_WG], s_sq[GAE_WG];
  const int b = blockIdx.x * GAE_WG + threadIdx.x;
  double sum = 0.0, sq = 0.0;
  if (b < B) {
    float advantage = 0.0f, next_v = last_values[b];
    for (int t = T - 1; t >= 0; --t) {
      const size_t i = (size_t)t * B + b;
      const float not_terminal = 1.0f - (float)don[i];
      const float v = val[i];
      const float delta = (rew[i] + (not_terminal * gamma) * next_v) - v;
      advantage = delta + ((not_terminal * gamma) * lam) * advantage;
      const float r = advantage + v;
      ret[i] = r;
      const float a = r - v;
      adv[i] = a;
      sum += (double)a; sq += (double)a * (double)a;
      next_v = v;
    }
  }
  s_sum[threadIdx.x] = sum; s_sq[threadIdx.x] = sq;
  __syncthreads();
  for (int s = GAE_WG / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s_sum[threadIdx.x] += s_sum[threadIdx.x + s]; s_sq[threadIdx.x] += s_sq[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_sum[0]; partial[2 * blockIdx.x + 1] = s_sq[0]; }
}
__global__ void k_gae_moments(const double* __restrict__ partial, int n_wg, double count, double* __restrict__ moments3) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < n_wg; ++i) { s += partial[2 * i]; q += partial[2 * i + 1]; }
  moments3[0] = s; moments3[1] = q; moments3[2] = count;
}
__global__ __launch_bounds__(256) void k_adv_normalize(float* __restrict__ adv, size_t n, const double* __restrict__ moments3) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double N = moments3[2], mean = moments3[0] / N;
  double var = (moments3[1] - N * mean * mean) / (N > 1.0 ? N - 1.0 : 1.0);   // torch.std: unbiased
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, stdf = (float)sqrt(var);
  adv[i] = (adv[i] - meanf) / (stdf + 1e-8f);
}

int round16(int v) { return (v + 15) / 16 * 16; }

}  // namespace

struct go2sim_mlp {
  int device = 0;
  int n_layers = 0;
  int dims[MAXL + 1] = {0};
  size_t n_params = 0;
  float* dparams = nullptr;   // padded weights + biases, one allocation
  size_t padded = 0;
  MlpDev dev{};
  float* scratch_mean = nullptr; int scratch_rows = 0;   // mean buffer of go2sim_policy_act when the caller passes mean == NULL
};

struct go2sim_rollout {
  int T = 0, B = 0, n_wg = 0;
  float* f = nullptr;        // rewards | values | returns | advantages, each [T][B]
  uint8_t* dones = nullptr;  // [T][B]
  double* partial = nullptr; // per-workgroup (sum, sum of squares)
};

namespace {
size_t expected_params(const int* dims, int n_layers) {
  size_t n = 0;
  for (int l = 0; l < n_layers; ++l) n += (size_t)dims[l] * dims[l + 1] + dims[l + 1];
  return n;
}
// pack [out][in] + [out] into the zero-padded device layout
void pack_padded(const go2sim_mlp* h, const float* params, std::vector<float>& out) {
  out.assign(h->padded, 0.0f);
  size_t src = 0, dst = 0;
  for (int l = 0; l < h->n_layers; ++l) {
    const int din = h->dims[l], dout = h->dims[l + 1], kp = round16(din), np = round16(dout);
    for (int n = 0; n < dout; ++n) memcpy(&out[dst + (size_t)n * kp], &params[src + (size_t)n * din], sizeof(float) * din);
    src += (size_t)din * dout; dst += (size_t)kp * np;
    memcpy(&out[dst], &params[src], sizeof(float) * dout);
    src += dout; dst += np;
  }
}
}  // namespace

extern "C" {

int go2sim_mlp_create(int device, const int* dims, int n_layers, const float* params, size_t n_params, go2sim_mlp_t** out) {
  if (!dims || !params || !out || n_layers < 1 || n_layers > MAXL) return GO2SIM_E_BADARG;
  for (int l = 0; l <= n_layers; ++l) if (dims[l] < 1 || dims[l] > MAXW) return GO2SIM_E_BADARG;
  if (n_params != expected_params(dims, n_layers)) return GO2SIM_E_BADARG;
  HIPCHK(hipSetDevice(device));
  go2sim_mlp* h = new (std::nothrow) go2sim_mlp();
  if (!h) return GO2SIM_E_NOMEM;
  h->device = device; h->n_layers = n_layers; h->n_params = n_params;
  memcpy(h->dims, dims, sizeof(int) * (n_layers + 1));
  size_t padded = 0;
  for (int l = 0; l < n_layers; ++l) padded += (size_t)round16(dims[l]) * round16(dims[l + 1]) + round16(dims[l + 1]);
  h->padded = padded;
  if (hipMalloc((void**)&h->dparams, padded * sizeof(float)) != hipSuccess) { delete h; return GO2SIM_E_NOMEM; }
  h->dev.n_layers = n_layers;
  size_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int kp = round16(dims[l]), np = round16(dims[l + 1]);
    h->dev.din[l] = dims[l]; h->dev.dout[l] = dims[l + 1]; h->dev.kpad[l] = kp; h->dev.npad[l] = np;
    h->dev.W[l] = h->dparams + off; off += (size_t)kp * np;
    h->dev.b[l] = h->dparams + off; off += np;
  }
  int rc = go2sim_mlp_set_params(h, params, n_params, nullptr);
  if (rc != GO2SIM_E_OK) { (void)hipFree(h->dparams); delete h; return rc; }
  HIPCHK(hipDeviceSynchronize());
  *out = h;
  return GO2SIM_E_OK;
}

int go2sim_mlp_destroy(go2sim_mlp_t* h) {
  if (!h) return GO2SIM_E_BADARG;
  (void)hipFree(h->dparams);
  if (h->scratch_mean) (void)hipFree(h->scratch_mean);
  delete h;
  return GO2SIM_E_OK;
}

int go2sim_mlp_set_params(go2sim_mlp_t* h, const float* params, size_t n_params, void* stream) {
  if (!h || !params || n_params != h->n_params) return GO2SIM_E_BADARG;
  std::vector<float> packed;
  pack_padded(h, params, packed);
  HIPCHK(hipMemcpyAsync(h->dparams, packed.data(), h->padded * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));   // `packed` is a temporary
  return GO2SIM_E_OK;
}

int go2sim_mlp_forward(go2sim_mlp_t* h, const float* x, float* y, int n_rows, void* stream) {
  if (!h || !x || !y || n_rows < 0) return GO2SIM_E_BADARG;
  if (n_rows == 0) return GO2SIM_E_OK;
  hipLaunchKernelGGL(k_mlp_forward, dim3((n_rows + TM - 1) / TM, 1), dim3(64 * NWAVE), 0, (hipStream_t)stream, h->dev, x, y, h->dev, x, y, n_rows);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}

int go2sim_policy_act(go2sim_mlp_t* actor, go2sim_mlp_t* critic, const float* obs, const float* critic_obs, const float* std_, int n_rows,
                      uint64_t seed, uint32_t step, int deterministic, float* actions, float* mean, float* values, float* log_prob, void* stream) {
  if (!actor || !obs || !std_ || !actions || n_rows < 0) return GO2SIM_E_BADARG;
  if ((values != nullptr) != (critic != nullptr && critic_obs != nullptr)) return GO2SIM_E_BADARG;
  if (critic && critic->dims[critic->n_layers] != 1) return GO2SIM_E_BADARG;
  if (n_rows == 0) return GO2SIM_E_OK;
  const int A = actor->dims[actor->n_layers];
  float* mu = mean;
  if (!mu) {
    if (actor->scratch_rows < n_rows) {
      if (actor->scratch_mean) (void)hipFree(actor->scratch_mean);
      actor->scratch_mean = nullptr; actor->scratch_rows = 0;
      if (hipMalloc((void**)&actor->scratch_mean, (size_t)n_rows * A * sizeof(float)) != hipSuccess) return GO2SIM_E_NOMEM;
      actor->scratch_rows = n_rows;
    }
    mu = actor->scratch_mean;
  }
  if (critic) {
    hipLaunchKernelGGL(k_mlp_forward, dim3((n_rows + TM - 1) / TM, 2), dim3(64 * NWAVE), 0, (hipStream_t)stream, actor->dev, obs, mu, critic->dev, critic_obs, values, n_rows);
    HIPCHK(hipGetLastError());
  } else {
    int rc = go2sim_mlp_forward(actor, obs, mu, n_rows, stream);
    if (rc != GO2SIM_E_OK) return rc;
  }
  hipLaunchKernelGGL(k_policy_sample, dim3((n_rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, mu, std_, n_rows, A, seed, step, deterministic, actions, log_prob);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}

// ---- rollout storage ----
int go2sim_rollout_create(int device, int n_steps, int n_envs, go2sim_rollout_t** out) {
  if (!out || n_steps < 1 || n_envs < 1) return GO2SIM_E_BADARG;
  HIPCHK(hipSetDevice(device));
  go2sim_rollout* h = new (std::nothrow) go2sim_rollout();
  if (!h) return GO2SIM_E_NOMEM;
  h->T = n_steps; h->B = n_envs; h->n_wg = (n_envs + GAE_WG - 1) / GAE_WG;
  const size_t n = (size_t)n_steps * n_envs;
  if (hipMalloc((void**)&h->f, 4 * n * sizeof(float)) != hipSuccess || hipMalloc((void**)&h->dones, n) != hipSuccess ||
      hipMalloc((void**)&h->partial, 2 * (size_t)h->n_wg * sizeof(double)) != hipSuccess) { go2sim_rollout_destroy(h); return GO2SIM_E_NOMEM; }
  HIPCHK(hipMemset(h->f, 0, 4 * n * sizeof(float)));
  HIPCHK(hipMemset(h->dones, 0, n));
  *out = h;
  return GO2SIM_E_OK;
}
int go2sim_rollout_destroy(go2sim_rollout_t* h) {
  if (!h) return GO2SIM_E_BADARG;
  if (h->f) (void)hipFree(h->f);
  if (h->dones) (void)hipFree(h->dones);
  if (h->partial) (void)hipFree(h->partial);
  delete h;
  return GO2SIM_E_OK;
}
int go2sim_rollout_add(go2sim_rollout_t* h, int t, const float* rewards, const uint8_t* dones, const float* values, const float* time_outs, float gamma, void* stream) {
  if (!h || t < 0 || t >= h->T || !rewards || !dones || !values) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B, o = (size_t)t * h->B;
  hipLaunchKernelGGL(k_rollout_add, dim3((h->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->f + o, h->f + n + o, h->dones + o, h->B, rewards, dones, values, time_outs, gamma);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_rollout_compute_returns(go2sim_rollout_t* h, const float* last_values, float gamma, float lam, double* moments3, void* stream) {
  if (!h || !last_values || !moments3) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B;
  hipLaunchKernelGGL(k_gae, dim3(h->n_wg), dim3(GAE_WG), 0, (hipStream_t)stream, h->f, h->f + n, h->dones, last_values, h->T, h->B, gamma, lam, h->f + 2 * n, h->f + 3 * n, h->partial);
  hipLaunchKernelGGL(k_gae_moments, dim3(1), dim3(1), 0, (hipStream_t)stream, h->partial, h->n_wg, (double)n, moments3);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_rollout_normalize(go2sim_rollout_t* h, const double* moments3, void* stream) {
  if (!h || !moments3) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B;
  hipLaunchKernelGGL(k_adv_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->f + 3 * n, n, moments3);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_rollout_ptr(go2sim_rollout_t* h, int buf, void** out) {
  if (!h || !out) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B;
  switch (buf) {
    case GO2SIM_RB_REWARDS: *out = h->f; break;
    case GO2SIM_RB_VALUES: *out = h->f + n; break;
    case GO2SIM_RB_DONES: *out = h->dones; break;
    case GO2SIM_RB_RETURNS: *out = h->f + 2 * n; break;
    case GO2SIM_RB_ADVANTAGES: *out = h->f + 3 * n; break;
    default: return GO2SIM_E_BADARG;
  }
  return GO2SIM_E_OK;
}

}  // extern "C"
