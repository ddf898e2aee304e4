// Reduced line-search kernel for the `update_bracket` question (DESIGN.md "CPU <-> GPU bit exactness").
//
// ts_linesearch (csrc/go2sim.hip; func_linesearch_batch, constraint/solver.py:2246-2417) refines a bracket with the 3-point step
// update_bracket_no_eval_local (solver.py:2212-2243).  In the full solver kernel the results differ from the CPU oracle when update_bracket is
// inlined; this file holds the same line search on plain arrays (one problem per lane) in two device builds -- bracket step inlined /
// out of line -- next to the host build of the same source.  It reports, for random problems, how many lanes differ from the host result.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/repro_bracket/ls_bracket_repro.hip -o tools/repro_bracket/ls_bracket_repro && ./ls_bracket_repro
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HD __host__ __device__
constexpr int NROW = 16, LS_IT = 50;
struct LsPoint { float alpha, cost, grad, hess; };
struct Problem { float Jaref[NROW], jv[NROW], D[NROW], qg[3], gtol, eps; };

template <bool INL> struct Br;
HD inline int bracket_body(LsPoint& p, const float* al, const float* c, const float* g, const float* h, float& next) {
  int flag = 0;
  for (int i = 0; i < 3; ++i) {
    if (p.grad < 0 && g[i] < 0 && p.grad < g[i]) { p.alpha = al[i]; p.cost = c[i]; p.grad = g[i]; p.hess = h[i]; flag = 1; }
    else if (p.grad > 0 && g[i] > 0 && p.grad > g[i]) { p.alpha = al[i]; p.cost = c[i]; p.grad = g[i]; p.hess = h[i]; flag = 2; }
  }
  next = p.alpha;
  if (flag > 0) next = p.alpha - p.grad / p.hess;
  return flag;
}
template <> struct Br<true> { HD static __forceinline__ int f(LsPoint& p, const float* al, const float* c, const float* g, const float* h, float& n) { return bracket_body(p, al, c, g, h, n); } };
template <> struct Br<false> { HD static __attribute__((noinline)) int f(LsPoint& p, const float* al, const float* c, const float* g, const float* h, float& n) { return bracket_body(p, al, c, g, h, n); } };

HD inline LsPoint point(const Problem& P, float alpha) {
  float t0 = P.qg[0], t1 = P.qg[1], t2 = P.qg[2];
  for (int c = 0; c < NROW; ++c) {
    float Ja = P.Jaref[c], jv = P.jv[c], D = P.D[c];
    float x = Ja + alpha * jv, act = (float)(x < 0.0f);
    t0 = t0 + D * (0.5f * Ja * Ja) * act; t1 = t1 + D * (jv * Ja) * act; t2 = t2 + D * (0.5f * jv * jv) * act;
  }
  LsPoint p; p.alpha = alpha; p.cost = alpha * alpha * t2 + alpha * t1 + t0; p.grad = 2.0f * alpha * t2 + t1; p.hess = 2.0f * t2;
  if (p.hess <= 0.0f) p.hess = P.eps;
  return p;
}
template <bool INL>
HD float linesearch(const Problem& P) {
  const float gtol = P.gtol;
  LsPoint p0 = point(P, 0.0f);
  LsPoint p1 = point(P, p0.alpha - p0.grad / p0.hess);
  int ls_it = 2;
  if (p0.cost < p1.cost) p1 = p0;
  if (fabsf(p1.grad) < gtol) return p1.alpha;
  int direction = (p1.grad < 0) * 2 - 1, p2update = 0;
  LsPoint p2 = p1;
  while (p1.grad * (float)direction <= -gtol && ls_it < LS_IT) {
    p2 = p1; p2update = 1;
    p1 = point(P, p1.alpha - p1.grad / p1.hess); ls_it += 1;
    if (fabsf(p1.grad) < gtol) return p1.alpha;
  }
  if (ls_it >= LS_IT || !p2update) return p1.alpha;
  float al[3] = {p1.alpha - p1.grad / p1.hess, p1.alpha, (p1.alpha + p2.alpha) * 0.5f};
  while (ls_it < LS_IT) {
    float c[3], g[3], h[3];
    for (int k = 0; k < 3; ++k) { LsPoint q = point(P, al[k]); c[k] = q.cost; g[k] = q.grad; h[k] = q.hess; }
    ls_it += 3;
    float n1 = al[0], n2 = al[1], best_a = 0.0f, best_c = 0.0f; bool found = false;
    for (int i = 0; i < 3; ++i) if (fabsf(g[i]) < gtol && (!found || c[i] < best_c)) { best_a = al[i]; best_c = c[i]; found = true; }
    if (found) return best_a;
    int b1 = Br<INL>::f(p1, al, c, g, h, n1), b2 = Br<INL>::f(p2, al, c, g, h, n2);
    if (b1 == 0 && b2 == 0) return al[2];
    al[0] = n1; al[1] = n2; al[2] = (p1.alpha + p2.alpha) * 0.5f;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0.0f;
}
template <bool INL> __global__ void k_ls(const Problem* P, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = linesearch<INL>(P[i]); }

int main() {
  const int N = 1 << 16;
  std::vector<Problem> h(N);
  srand(12345);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
  for (auto& p : h) {
    for (int c = 0; c < NROW; ++c) { p.Jaref[c] = 4.0f * rnd() - 2.5f; p.jv[c] = 8.0f * rnd() - 4.0f; p.D[c] = 50.0f + 3000.0f * rnd(); }
    p.qg[0] = 10.0f * rnd(); p.qg[1] = 40.0f * rnd() - 20.0f; p.qg[2] = 0.5f + 20.0f * rnd(); p.gtol = 1e-6f * (1.0f + 100.0f * rnd()); p.eps = 1e-15f;
  }
  Problem* d; float* o;
  hipMalloc(&d, N * sizeof(Problem)); hipMalloc(&o, N * sizeof(float));
  hipMemcpy(d, h.data(), N * sizeof(Problem), hipMemcpyHostToDevice);
  std::vector<float> ref(N), got(N);
  for (int i = 0; i < N; ++i) ref[i] = linesearch<true>(h[i]);
  int bad[2] = {0, 0};
  for (int v = 0; v < 2; ++v) {
    if (v == 0) hipLaunchKernelGGL(k_ls<true>, dim3(N / 64), dim3(64), 0, 0, d, o, N); else hipLaunchKernelGGL(k_ls<false>, dim3(N / 64), dim3(64), 0, 0, d, o, N);
    hipMemcpy(got.data(), o, N * sizeof(float), hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) { unsigned a, b; memcpy(&a, &ref[i], 4); memcpy(&b, &got[i], 4); if (a != b) { if (bad[v] < 3) printf("  variant %d lane %d host %.9g device %.9g\n", v, i, ref[i], got[i]); bad[v]++; } }
  }
  printf("line-search problems: %d; lanes differing from the host build: bracket inlined %d, bracket out of line %d\n", N, bad[0], bad[1]);
  return (bad[0] || bad[1]) ? 1 : 0;
}
