/* go2sim_policy.h -- C ABI of the policy-inference step that sits next to the environment step in the rollout loop
 * ("what comes next" after the env hot path).
 *
 * What it replaces on the reference side: `rsl_rl.modules.ActorCritic` as the train scripts configure it
 * (examples/locomotion/final/go2_train_walk.py:41-47: activation "elu", actor_hidden_dims = critic_hidden_dims = [512, 256, 128],
 * init_noise_std 1.0; rsl-rl-lib==2.2.4 is a third-party dependency that is NOT part of /root/reference -- go2_train_walk.py:12-15 pins
 * the version).  The calls OnPolicyRunner / PPO make on it per environment step are
 *     actions  = ActorCritic.act(obs)            -> mean = actor(obs); Normal(mean, std).sample()
 *     values   = ActorCritic.evaluate(critic_obs)
 *     log_prob = ActorCritic.get_actions_log_prob(actions)
 * with `actor` / `critic` = nn.Sequential(Linear, ELU, Linear, ELU, Linear, ELU, Linear) and `std` a learned vector.
 *
 * Conventions follow include/go2sim.h: extern "C", plain pointers and sizes, int status (0 = ok, GO2SIM_E_*), device pointers are
 * caller-owned (torch storage), work is ordered on the given stream, a handle is not thread-safe.  The CPU oracle exports the same
 * functions with the prefix go2sim_cpu_ and host pointers (test infrastructure only).
 *
 * Numerics: fp32 throughout.  A linear layer is evaluated as the k-ordered fp32 fused-multiply-add chain of the gfx950 fp32 matrix
 * instruction (v_mfma_f32_16x16x4_f32) with the K index visited in the order documented at go2sim_mlp_forward; the CPU oracle uses the
 * same order with fmaf, so the two agree bit for bit; against a PyTorch fp32 nn.Sequential the difference is summation order only
 * (tests/test_policy.py: |diff| <= 2e-5 + 2e-5 |ref| at unit-scale activations).
 */
#ifndef GO2SIM_POLICY_H
#define GO2SIM_POLICY_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO2SIM_MLP_MAX_LAYERS 6
#define GO2SIM_MLP_MAX_WIDTH 512

typedef struct go2sim_mlp go2sim_mlp_t;

/* One multilayer perceptron: n_layers Linear layers of sizes dims[0] -> dims[1] -> ... -> dims[n_layers], ELU (alpha 1) after every layer but
 * the last (rsl_rl ActorCritic with activation="elu").  `params` (host memory) holds, layer by layer, the weight matrix in torch's
 * Linear.weight layout [out][in] row-major followed by the bias [out] -- i.e. the tensors of the state dict in order.
 * Widths up to GO2SIM_MLP_MAX_WIDTH, up to GO2SIM_MLP_MAX_LAYERS layers.  `device` is the HIP device index (ignored by the CPU twin). */
int go2sim_mlp_create(int device, const int* dims, int n_layers, const float* params, size_t n_params, go2sim_mlp_t** out);
int go2sim_mlp_destroy(go2sim_mlp_t* h);
/* Replaces the parameters (same shapes), e.g. after an optimizer step of the caller. */
int go2sim_mlp_set_params(go2sim_mlp_t* h, const float* params, size_t n_params, void* stream);

/* y[B][dims[n_layers]] = mlp(x[B][dims[0]]); row-major, device pointers.  One kernel launch: a workgroup keeps the activations of 16 rows
 * in LDS across all layers.  Summation order of one output element: acc = 0; for K blocks of 16 (zero-padded): for s in 0..3: for q in 0..3:
 * k = 16 * block + 4 * q + s: acc = fma(x[k], W[n][k], acc); then + bias, then ELU. */
int go2sim_mlp_forward(go2sim_mlp_t* h, const float* x, float* y, int n_rows, void* stream);

/* ActorCritic.act + evaluate + get_actions_log_prob for one environment step:
 *   mean[B][A]    = actor(obs[B][n_obs])
 *   values[B]     = critic(critic_obs[B][n_critic_obs])
 *   actions[B][A] = mean + std[A] * n,   n ~ N(0,1) from the counter-based Philox stream of include/go2sim_detmath.h keyed
 *                   (seed; row, step, purpose 11, block) -- the library's replacement for torch's global generator (torch.normal)
 *   log_prob[B]   = sum_a ( -((actions - mean)^2) / (2 std^2) - log(std) - log(sqrt(2 pi)) )      (torch.distributions.Normal.log_prob)
 * Any of mean / values / log_prob may be NULL.  critic may be NULL (then values must be NULL).  deterministic != 0 gives
 * ActorCritic.act_inference (actions = mean). */
int go2sim_policy_act(go2sim_mlp_t* actor, go2sim_mlp_t* critic, const float* obs, const float* critic_obs, const float* std, int n_rows,
                      uint64_t seed, uint32_t step, int deterministic, float* actions, float* mean, float* values, float* log_prob, void* stream);

/* ---- rollout storage: returns / advantages of one rollout on the device ----------------------------------------
 * Replaces rsl_rl.storage.RolloutStorage.add_transitions / compute_returns as PPO uses them (rsl-rl-lib==2.2.4, third party):
 *   PPO.process_env_step : rewards += gamma * values * time_outs            (bootstrap on time-outs)
 *   compute_returns      : advantage = 0; for t = T-1 .. 0:
 *                            not_terminal = 1 - dones[t];  next_v = (t == T-1) ? last_values : values[t+1]
 *                            delta = rewards[t] + not_terminal * gamma * next_v - values[t]
 *                            advantage = delta + not_terminal * gamma * lam * advantage;  returns[t] = advantage + values[t]
 *                          advantages = returns - values;  advantages = (advantages - mean) / (std + 1e-8)   (torch.std: unbiased)
 * All arrays are [T][B] row-major on the device, owned by the handle.  The mean / std are taken over ALL ranks of a multi-GPU job: the
 * library produces the local moments [sum, sum of squares, count] (float64, fixed summation order => bit-reproducible), the caller
 * all-gathers them over RCCL (distributed.py) and hands the global moments to go2sim_rollout_normalize. */
typedef struct go2sim_rollout go2sim_rollout_t;
enum go2sim_rollout_buf { GO2SIM_RB_REWARDS = 0, GO2SIM_RB_VALUES, GO2SIM_RB_DONES /* u8 */, GO2SIM_RB_RETURNS, GO2SIM_RB_ADVANTAGES };
int go2sim_rollout_create(int device, int n_steps, int n_envs, go2sim_rollout_t** out);
int go2sim_rollout_destroy(go2sim_rollout_t* h);
/* transition t of the rollout (device pointers [n_envs]); time_outs may be NULL */
int go2sim_rollout_add(go2sim_rollout_t* h, int t, const float* rewards, const uint8_t* dones, const float* values, const float* time_outs, float gamma, void* stream);
/* returns, un-normalised advantages and the local moments (moments3: device pointer to 3 float64) */
int go2sim_rollout_compute_returns(go2sim_rollout_t* h, const float* last_values, float gamma, float lam, double* moments3, void* stream);
/* advantages <- (advantages - mean) / (std + 1e-8) with the (global) moments */
int go2sim_rollout_normalize(go2sim_rollout_t* h, const double* moments3, void* stream);
/* zero-copy device pointer of one buffer (no ownership transfer) */
int go2sim_rollout_ptr(go2sim_rollout_t* h, int buf, void** out);

#ifdef __cplusplus
}
#endif
#endif
