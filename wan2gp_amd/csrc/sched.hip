// wan_sched_*: the flow-matching sampler step behind the C ABI.
//
//   kind 0  UniPC  -- shared/utils/fm_solvers_unipc.py, the configuration WanAny2V.generate() builds (any2video.py:520-523):
//                     bh2, solver_order 2, predict_x0, flow_prediction, lower_order_final, final sigma 0, constructor shift 1
//   kind 1  Euler  -- shared/utils/euler_scheduler.py:26-87 (use_timestep_transform)
//
// The reference does the scalar algebra of a step on the host (fp32 torch scalars) and the tensor updates as a chain of eager
// fp32 ops over the latents.  Here the scalar algebra is host C++ in the same precision and the same order of operations
// (set_timesteps in double like numpy, the per-step coefficients in float like torch's fp32 scalars); every tensor update is
// linear in (sample, last corrected sample, x0 predictions) with host-known coefficients and collapses into ONE wan_lincomb
// launch.  The x0-prediction history and the corrected sample live in library-owned fp32 buffers (5 x n floats, allocated at
// the first step), so a step is: 1 launch (x0) + 1 (UniC, from the second step on) + 1 (UniP).
#include <cmath>
#include <vector>

#include "common.h"

extern "C" int wan_lincomb(float* out, int n_in, const float* const* in, const float* coef, int64_t n, void* stream);

struct wan_sched {
  int kind = 0;
  int ntrain = 1000;
  int nsteps = 0;
  std::vector<float> sigmas;     // UniPC: nsteps + 1 (last = 0)
  std::vector<double> tsteps;    // UniPC: int64 values; Euler: fp32 values
  // UniPC state (fm_solvers_unipc.py:147-153)
  int step_index = -1, lower_order_nums = 0, this_order = 1, n_hist = 0;
  bool have_last = false;
  float* m[3] = {nullptr, nullptr, nullptr};  // x0 predictions: m[0] older, m[1] newest, m[2] scratch for this step's
  float* last[2] = {nullptr, nullptr};        // corrected sample of the previous step / scratch
  int64_t cap = 0;
};

namespace {

inline float lam(float sigma) { return logf(1.0f - sigma) - logf(sigma); }

// multistep_uni_p/c_bh_update's scalar part (fm_solvers_unipc.py:405-449 / :545-590) for orders 1 and 2, solver bh2
struct Bh {
  float rk0, b[2], h_phi_1, B_h;
};
inline Bh bh(float sig_t, float sig_s0, int order, float sig_prev) {
  Bh r;
  const float lam_t = lam(sig_t), lam_s0 = lam(sig_s0);
  const float h = lam_t - lam_s0;
  r.rk0 = (order == 2) ? (lam(sig_prev) - lam_s0) / h : 1.0f;
  const float hh = -h;
  r.h_phi_1 = expm1f(hh);
  float h_phi_k = r.h_phi_1 / hh - 1.0f;
  r.B_h = expm1f(hh);
  float fact = 1.0f;
  for (int i = 1; i <= order; ++i) {
    r.b[i - 1] = h_phi_k * fact / r.B_h;
    fact *= (float)(i + 1);
    h_phi_k = h_phi_k / hh - 1.0f / fact;
  }
  return r;
}

int ensure(wan_sched* s, int64_t n) {
  if (n <= s->cap) return 0;
  float** all[5] = {&s->m[0], &s->m[1], &s->m[2], &s->last[0], &s->last[1]};
  for (auto p : all) {
    if (*p) WAN_CHECK_HIP(hipFree(*p));
    *p = nullptr;
    WAN_CHECK_HIP(hipMalloc((void**)p, (size_t)n * sizeof(float)));
  }
  s->cap = n;
  return 0;
}

}  // namespace

extern "C" int wan_sched_create(wan_sched** out, int kind, int num_train_timesteps) {
  WAN_REQUIRE(out != nullptr, "wan_sched_create: null out");
  WAN_REQUIRE(kind == 0 || kind == 1, "wan_sched_create: kind %d (0 = UniPC, 1 = Euler)", kind);
  WAN_REQUIRE(num_train_timesteps >= 2, "wan_sched_create: num_train_timesteps=%d", num_train_timesteps);
  wan_sched* s = new wan_sched();
  s->kind = kind;
  s->ntrain = num_train_timesteps;
  *out = s;
  return 0;
}

extern "C" void wan_sched_destroy(wan_sched* s) {
  if (!s) return;
  for (float* p : {s->m[0], s->m[1], s->m[2], s->last[0], s->last[1]})
    if (p) (void)hipFree(p);
  delete s;
}

extern "C" int wan_sched_set_timesteps(wan_sched* s, int steps, double shift, double* timesteps_out, float* sigmas_out) {
  WAN_REQUIRE(s != nullptr && steps >= 1, "wan_sched_set_timesteps: bad arguments (steps=%d)", steps);
  s->nsteps = steps;
  s->tsteps.assign(steps, 0.0);
  if (s->kind == 0) {
    // __init__ (:134-145, shift 1): sigmas = 1 - linspace(1, 1/N, N)[::-1] as fp32 -> sigma_max = fp32(1 - 1/N), sigma_min = 0
    const int N = s->ntrain;
    const double sigma_max = (double)(float)(1.0 - 1.0 / (double)N), sigma_min = 0.0;  // linspace's last element is `stop` itself
    // set_timesteps (:186-215): linspace(sigma_max, sigma_min, steps + 1)[:-1] in double, shifted, timesteps truncated to int64
    s->sigmas.assign(steps + 1, 0.f);
    for (int i = 0; i < steps; ++i) {
      double sg = sigma_max + (double)i * ((sigma_min - sigma_max) / (double)steps);
      sg = shift * sg / (1.0 + (shift - 1.0) * sg);
      s->sigmas[i] = (float)sg;
      s->tsteps[i] = (double)(int64_t)(sg * (double)N);
    }
    s->sigmas[steps] = 0.f;
  } else {
    // euler_scheduler.py:52-66: linspace(N, 1, steps) as fp32, then t/N -> shift*t/(1+(shift-1)*t)*N in fp32
    s->sigmas.clear();
    const float sh = (float)shift, shm1 = (float)(shift - 1.0);
    for (int i = 0; i < steps; ++i) {
      const double lin = (steps == 1) ? (double)s->ntrain : (double)s->ntrain + (double)i * ((1.0 - (double)s->ntrain) / (double)(steps - 1));
      float t = (float)lin;
      t = t / (float)s->ntrain;
      t = sh * t / (1.0f + shm1 * t) * (float)s->ntrain;
      s->tsteps[i] = (double)t;
    }
  }
  s->step_index = -1;
  s->lower_order_nums = 0;
  s->this_order = 1;
  s->n_hist = 0;
  s->have_last = false;
  if (timesteps_out)
    for (int i = 0; i < steps; ++i) timesteps_out[i] = s->tsteps[i];
  if (sigmas_out && s->kind == 0)
    for (int i = 0; i <= steps; ++i) sigmas_out[i] = s->sigmas[i];
  return 0;
}

extern "C" int wan_sched_step(wan_sched* s, const float* model_output, double timestep, const float* sample, float* prev_out,
                              int64_t n, void* stream) {
  WAN_REQUIRE(s && model_output && sample && prev_out && n > 0, "wan_sched_step: null pointer / n <= 0");
  WAN_REQUIRE(s->nsteps > 0, "wan_sched_step: set_timesteps has not been called");
  WAN_REQUIRE(prev_out != sample && prev_out != model_output, "wan_sched_step: prev_out must not alias an input");
  if (s->kind == 1) {  // euler_scheduler.py:68-87
    int idx = 0;
    double best = fabs((double)(float)s->tsteps[0] - (double)(float)timestep);
    for (int i = 1; i < s->nsteps; ++i) {
      const float dlt = fabsf((float)s->tsteps[i] - (float)timestep);
      if ((double)dlt < best) { best = dlt; idx = i; }
    }
    const float dt_raw = (idx + 1 < s->nsteps) ? (float)s->tsteps[idx] - (float)s->tsteps[idx + 1] : (float)s->tsteps[idx];
    const double dt = (double)dt_raw / (double)s->ntrain;
    const float* in[2] = {sample, model_output};
    const float cf[2] = {1.0f, (float)-dt};
    return wan_lincomb(prev_out, 2, in, cf, n, stream);
  }
  // ---- UniPC (fm_solvers_unipc.py:640-721) ----
  if (int rc = ensure(s, n)) return rc;
  if (s->step_index < 0) {  // index_for_timestep (:630-637): the second match if the timestep occurs twice
    int first = -1, second = -1;
    for (int i = 0; i < s->nsteps; ++i)
      if (s->tsteps[i] == (double)(int64_t)timestep) {
        if (first < 0) first = i; else if (second < 0) second = i;
      }
    WAN_REQUIRE(first >= 0, "wan_sched_step: timestep %.1f is not one of the scheduler's", timestep);
    s->step_index = second >= 0 ? second : first;
  }
  const int i = s->step_index;
  WAN_REQUIRE(i < s->nsteps, "wan_sched_step: stepped past the last timestep");
  const std::vector<float>& sig = s->sigmas;
  // x0 = x - sigma * v  (:313-315)
  float* m_t = s->m[2];
  {
    const float* in[2] = {sample, model_output};
    const float cf[2] = {1.0f, -sig[i]};
    if (int rc = wan_lincomb(m_t, 2, in, cf, n, stream)) return rc;
  }
  const float* cur = sample;  // the sample the predictor starts from: corrected below when a corrector step applies
  if (i > 0 && s->have_last) {  // UniC (:482-626)
    const int order = s->this_order;
    const float sig_t = sig[i], sig_s0 = sig[i - 1];
    const Bh c = bh(sig_t, sig_s0, order, order == 2 ? sig[i - 2] : 0.f);
    const float alpha_t = 1.0f - sig_t;
    const float c_last = sig_t / sig_s0;
    float c_m0, c_mt, c_m1 = 0.f;
    if (order == 1) {
      const float rho_last = 0.5f;
      c_m0 = -alpha_t * c.h_phi_1 + alpha_t * c.B_h * rho_last;
      c_mt = -alpha_t * c.B_h * rho_last;
    } else {  // R = [[1, 1], [rk0, 1]], R rhos = b
      const float det = 1.0f - c.rk0;
      const float rho0 = (c.b[0] - c.b[1]) / det, rho_last = (c.b[1] - c.rk0 * c.b[0]) / det;
      c_m1 = -alpha_t * c.B_h * rho0 / c.rk0;
      c_m0 = -alpha_t * c.h_phi_1 + alpha_t * c.B_h * (rho0 / c.rk0 + rho_last);
      c_mt = -alpha_t * c.B_h * rho_last;
    }
    const float* in[4] = {s->last[0], s->m[1], m_t, s->m[0]};
    const float cf[4] = {c_last, c_m0, c_mt, c_m1};
    if (int rc = wan_lincomb(s->last[1], order == 2 ? 4 : 3, in, cf, n, stream)) return rc;
    std::swap(s->last[0], s->last[1]);
    cur = s->last[0];
  } else {
    WAN_CHECK_HIP(hipMemcpyAsync(s->last[0], sample, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
  }
  s->have_last = true;
  // history shift: model_outputs[-2] <- model_outputs[-1] <- m_t
  {
    float* old = s->m[0];
    s->m[0] = s->m[1];
    s->m[1] = m_t;
    s->m[2] = old;
  }
  int this_order = std::min(2, s->nsteps - i);  // lower_order_final
  s->this_order = std::min(this_order, s->lower_order_nums + 1);
  // UniP (:350-480)
  {
    const int order = s->this_order;
    const float sig_t = sig[i + 1], sig_s0 = sig[i];
    const Bh p = bh(sig_t, sig_s0, order, order == 2 ? sig[i - 1] : 0.f);
    const float alpha_t = 1.0f - sig_t;
    const float c_x = sig_t / sig_s0;
    if (order == 2) {
      const float c_m1 = -alpha_t * p.B_h * 0.5f / p.rk0;
      const float c_m0 = -alpha_t * p.h_phi_1 + alpha_t * p.B_h * 0.5f / p.rk0;
      const float* in[3] = {cur, s->m[1], s->m[0]};
      const float cf[3] = {c_x, c_m0, c_m1};
      if (int rc = wan_lincomb(prev_out, 3, in, cf, n, stream)) return rc;
    } else {
      const float* in[2] = {cur, s->m[1]};
      const float cf[2] = {c_x, -alpha_t * p.h_phi_1};
      if (int rc = wan_lincomb(prev_out, 2, in, cf, n, stream)) return rc;
    }
  }
  if (s->lower_order_nums < 2) s->lower_order_nums += 1;
  s->step_index += 1;
  return 0;
}
