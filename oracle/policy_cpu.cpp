// policy_cpu.cpp -- CPU oracle of include/go2sim_policy.h (prefix go2sim_cpu_).  TEST INFRASTRUCTURE ONLY: imported by tests/,
// __graft_entry__.smoke() and nothing else; the product path is csrc/go2sim_policy.hip and has no CPU fallback.
//
// What it restates: rsl_rl.modules.ActorCritic (rsl-rl-lib==2.2.4, pinned by examples/locomotion/final/go2_train_walk.py:12-15; a
// third-party dependency that is not part of /root/reference) as configured by go2_train_walk.py:41-47:
//   actor / critic = nn.Sequential(Linear, ELU, Linear, ELU, Linear, ELU, Linear); act(): Normal(actor(obs), std).sample();
//   evaluate(): critic(critic_obs); get_actions_log_prob(): Normal.log_prob(actions).sum(-1).
// Parity: UNPINNED against rsl_rl itself (not importable here, no fixtures in the reference).  The restatement is pinned against a plain
// PyTorch fp32 nn.Sequential of the same architecture instead (tests/test_policy.py), which is the published definition of those modules.
// Linear layers use the summation order of the HIP kernel (see go2sim_policy.h) with fmaf, so GPU and oracle agree bit for bit.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../include/go2sim.h"
#include "../include/go2sim_detmath.h"
#include "../include/go2sim_policy.h"

struct go2sim_mlp {
  int n_layers = 0;
  int dims[GO2SIM_MLP_MAX_LAYERS + 1] = {0};
  std::vector<float> params;
};

struct go2sim_rollout { int T = 0, B = 0; std::vector<float> f; std::vector<uint8_t> dones; };

namespace {
size_t expected_params(const int* dims, int n_layers) {
  size_t n = 0;
  for (int l = 0; l < n_layers; ++l) n += (size_t)dims[l] * dims[l + 1] + dims[l + 1];
  return n;
}
inline float elu1(float v) { return v > 0.0f ? v : dm_exp(v) - 1.0f; }
// one row through the MLP
void forward_row(const go2sim_mlp* h, const float* x, float* y) {
  float a[GO2SIM_MLP_MAX_WIDTH + 16], o[GO2SIM_MLP_MAX_WIDTH + 16];
  for (int k = 0; k < h->dims[0]; ++k) a[k] = x[k];
  const float* p = h->params.data();
  for (int l = 0; l < h->n_layers; ++l) {
    const int din = h->dims[l], dout = h->dims[l + 1], kpad = (din + 15) / 16 * 16;
    for (int k = din; k < kpad; ++k) a[k] = 0.0f;
    const float* W = p; const float* b = p + (size_t)din * dout;
    for (int n = 0; n < dout; ++n) {
      float acc = 0.0f;
      for (int j = 0; j < kpad; j += 16)
        for (int s = 0; s < 4; ++s)
          for (int q = 0; q < 4; ++q) {
            const int k = j + 4 * q + s;
            const float w = (k < din) ? W[(size_t)n * din + k] : 0.0f;
            acc = fmaf(a[k], w, acc);
          }
      float v = acc + b[n];
      o[n] = (l == h->n_layers - 1) ? v : elu1(v);
    }
    for (int n = 0; n < dout; ++n) a[n] = o[n];
    p += (size_t)din * dout + dout;
  }
  for (int n = 0; n < h->dims[h->n_layers]; ++n) y[n] = a[n];
}
}  // namespace

extern "C" {

int go2sim_cpu_mlp_create(int, const int* dims, int n_layers, const float* params, size_t n_params, go2sim_mlp_t** out) {
  if (!dims || !params || !out || n_layers < 1 || n_layers > GO2SIM_MLP_MAX_LAYERS) return GO2SIM_E_BADARG;
  for (int l = 0; l <= n_layers; ++l) if (dims[l] < 1 || dims[l] > GO2SIM_MLP_MAX_WIDTH) return GO2SIM_E_BADARG;
  if (n_params != expected_params(dims, n_layers)) return GO2SIM_E_BADARG;
  go2sim_mlp* h = new (std::nothrow) go2sim_mlp();
  if (!h) return GO2SIM_E_NOMEM;
  h->n_layers = n_layers;
  memcpy(h->dims, dims, sizeof(int) * (n_layers + 1));
  h->params.assign(params, params + n_params);
  *out = h;
  return GO2SIM_E_OK;
}
int go2sim_cpu_mlp_destroy(go2sim_mlp_t* h) { if (!h) return GO2SIM_E_BADARG; delete h; return GO2SIM_E_OK; }
int go2sim_cpu_mlp_set_params(go2sim_mlp_t* h, const float* params, size_t n_params, void*) {
  if (!h || !params || n_params != h->params.size()) return GO2SIM_E_BADARG;
  h->params.assign(params, params + n_params);
  return GO2SIM_E_OK;
}
int go2sim_cpu_mlp_forward(go2sim_mlp_t* h, const float* x, float* y, int n_rows, void*) {
  if (!h || !x || !y || n_rows < 0) return GO2SIM_E_BADARG;
  const int din = h->dims[0], dout = h->dims[h->n_layers];
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n_rows; ++r) forward_row(h, x + (size_t)r * din, y + (size_t)r * dout);
  return GO2SIM_E_OK;
}
int go2sim_cpu_policy_act(go2sim_mlp_t* actor, go2sim_mlp_t* critic, const float* obs, const float* critic_obs, const float* std_, int n_rows,
                          uint64_t seed, uint32_t step, int deterministic, float* actions, float* mean, float* values, float* log_prob, void*) {
  if (!actor || !obs || !std_ || !actions || n_rows < 0) return GO2SIM_E_BADARG;
  if ((values != nullptr) != (critic != nullptr && critic_obs != nullptr)) return GO2SIM_E_BADARG;
  if (critic && critic->dims[critic->n_layers] != 1) return GO2SIM_E_BADARG;
  const int A = actor->dims[actor->n_layers];
  std::vector<float> mu_buf;
  float* mu = mean;
  if (!mu) { mu_buf.resize((size_t)n_rows * A); mu = mu_buf.data(); }
  go2sim_cpu_mlp_forward(actor, obs, mu, n_rows, nullptr);
  if (critic) go2sim_cpu_mlp_forward(critic, critic_obs, values, n_rows, nullptr);
  for (int b = 0; b < n_rows; ++b) {
    float lp = 0.0f;
    for (int blk = 0; 4 * blk < A; ++blk) {
      float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!deterministic) {
        dm_u4 r = dm_philox((uint32_t)b, step, 11u, (uint32_t)blk, (uint32_t)seed, (uint32_t)(seed >> 32));
        dm_normal2(r.v[0], r.v[1], &n[0], &n[1]); dm_normal2(r.v[2], r.v[3], &n[2], &n[3]);
      }
      for (int k = 0; k < 4; ++k) {
        const int a = 4 * blk + k;
        if (a >= A) break;
        const float m = mu[(size_t)b * A + a], sd = std_[a];
        const float act = m + sd * n[k];
        actions[(size_t)b * A + a] = act;
        const float d = act - m;
        lp = lp + ((-(d * d) / (2.0f * (sd * sd)) - dm_log(sd)) - 0.91893853320467274178f);
      }
    }
    if (log_prob) log_prob[b] = lp;
  }
  return GO2SIM_E_OK;
}

// ---- rollout storage (rsl_rl.storage.RolloutStorage.add_transitions / compute_returns + PPO.process_env_step; rsl-rl-lib==2.2.4) ----
int go2sim_cpu_rollout_create(int, int n_steps, int n_envs, go2sim_rollout_t** out) {
  if (!out || n_steps < 1 || n_envs < 1) return GO2SIM_E_BADARG;
  go2sim_rollout* h = new (std::nothrow) go2sim_rollout();
  if (!h) return GO2SIM_E_NOMEM;
  h->T = n_steps; h->B = n_envs;
  const size_t n = (size_t)n_steps * n_envs;
  h->f.assign(4 * n, 0.0f); h->dones.assign(n, 0);
  *out = h;
  return GO2SIM_E_OK;
}
int go2sim_cpu_rollout_destroy(go2sim_rollout_t* h) { if (!h) return GO2SIM_E_BADARG; delete h; return GO2SIM_E_OK; }
int go2sim_cpu_rollout_add(go2sim_rollout_t* h, int t, const float* rewards, const uint8_t* dones, const float* values, const float* time_outs, float gamma, void*) {
  if (!h || t < 0 || t >= h->T || !rewards || !dones || !values) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B, o = (size_t)t * h->B;
  for (int b = 0; b < h->B; ++b) {
    float rr = rewards[b];
    if (time_outs) rr = rr + gamma * (values[b] * time_outs[b]);
    h->f[o + b] = rr; h->f[n + o + b] = values[b]; h->dones[o + b] = dones[b];
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_rollout_compute_returns(go2sim_rollout_t* h, const float* last_values, float gamma, float lam, double* moments3, void*) {
  if (!h || !last_values || !moments3) return GO2SIM_E_BADARG;
  const int T = h->T, B = h->B;
  const size_t n = (size_t)T * B;
  const float* rew = h->f.data(); const float* val = rew + n; float* ret = h->f.data() + 2 * n; float* adv = h->f.data() + 3 * n;
  // same summation tree as the device kernel: 256-env workgroups, pairwise tree inside, workgroups in order
  const int WG = 256, n_wg = (B + WG - 1) / WG;
  double S = 0.0, Q = 0.0;
  for (int w = 0; w < n_wg; ++w) {
    double s_sum[256], s_sq[256];
    for (int l = 0; l < WG; ++l) {
      const int b = w * WG + l;
      double sum = 0.0, sq = 0.0;
      if (b < B) {
        float advantage = 0.0f, next_v = last_values[b];
        for (int t = T - 1; t >= 0; --t) {
          const size_t i = (size_t)t * B + b;
          const float not_terminal = 1.0f - (float)h->dones[i];
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
      s_sum[l] = sum; s_sq[l] = sq;
    }
    for (int s = WG / 2; s > 0; s >>= 1) for (int l = 0; l < s; ++l) { s_sum[l] += s_sum[l + s]; s_sq[l] += s_sq[l + s]; }
    S += s_sum[0]; Q += s_sq[0];
  }
  moments3[0] = S; moments3[1] = Q; moments3[2] = (double)n;
  return GO2SIM_E_OK;
}
int go2sim_cpu_rollout_normalize(go2sim_rollout_t* h, const double* moments3, void*) {
  if (!h || !moments3) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B;
  float* adv = h->f.data() + 3 * n;
  const double N = moments3[2], mean = moments3[0] / N;
  double var = (moments3[1] - N * mean * mean) / (N > 1.0 ? N - 1.0 : 1.0);
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, stdf = (float)sqrt(var);
  for (size_t i = 0; i < n; ++i) adv[i] = (adv[i] - meanf) / (stdf + 1e-8f);
  return GO2SIM_E_OK;
}
int go2sim_cpu_rollout_ptr(go2sim_rollout_t* h, int buf, void** out) {
  if (!h || !out) return GO2SIM_E_BADARG;
  const size_t n = (size_t)h->T * h->B;
  switch (buf) {
    case GO2SIM_RB_REWARDS: *out = h->f.data(); break;
    case GO2SIM_RB_VALUES: *out = h->f.data() + n; break;
    case GO2SIM_RB_DONES: *out = h->dones.data(); break;
    case GO2SIM_RB_RETURNS: *out = h->f.data() + 2 * n; break;
    case GO2SIM_RB_ADVANTAGES: *out = h->f.data() + 3 * n; break;
    default: return GO2SIM_E_BADARG;
  }
  return GO2SIM_E_OK;
}

}  // extern "C"
