"""Flow-matching schedulers with the reference's class/method surface, stepping on the GPU.

`FlowUniPCMultistepScheduler` mirrors shared/utils/fm_solvers_unipc.py (the default
`unipc` sampler, wan_handler.py:1162-1163) and `EulerScheduler` mirrors
shared/utils/euler_scheduler.py.  The reference does the (tiny) scalar algebra of each step
on the host and the tensor updates as a chain of eager fp32 ops over the latents; here the
scalar algebra is the same host code path (fp32 torch scalars, same expressions, same
order) and each tensor update collapses into ONE fused linear-combination kernel
(`wan_lincomb`) over the fp32 latents -- every UniP / UniC / x0 / Euler update is linear
in (sample, last_sample, model outputs) with host-known coefficients.
"""
import numpy as np
import torch

from . import ops


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample

    def __getitem__(self, i):
        return (self.prev_sample,)[i]


class FlowUniPCMultistepScheduler:
    """bh2 / order-2 / predict_x0 / flow_prediction / lower_order_final UniPC
    (fm_solvers_unipc.py:77-132 defaults, the only configuration generate() builds,
    any2video.py:520-523)."""
    order = 1

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", shift=1.0,
                 use_dynamic_shifting=False, predict_x0=True, solver_type="bh2", lower_order_final=True,
                 disable_corrector=(), final_sigmas_type="zero", **unused):
        if prediction_type != "flow_prediction" or not predict_x0 or solver_type != "bh2" or use_dynamic_shifting \
                or solver_order != 2 or final_sigmas_type != "zero":
            raise NotImplementedError("only the configuration WanAny2V.generate() uses is implemented")
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.shift = shift
        self.lower_order_final = lower_order_final
        self.disable_corrector = list(disable_corrector)
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.num_inference_steps = None
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None
        self.this_order = 1

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]   # :186-188
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                                            # :196-197
        timesteps = sigmas * self.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)                                       # :210-211
        self.sigmas = torch.from_numpy(sigmas)                                                          # stays on host
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)              # :214-215
        self._timesteps_host = self.timesteps.cpu()
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None

    # ---- scalar algebra (host, fp32 torch scalars exactly as fm_solvers_unipc.py:405-449 / :545-590) ----
    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _bh(self, sig_t, sig_s0, order, sig_prev):
        lam_t, lam_s0 = self._lam(sig_t), self._lam(sig_s0)
        h = lam_t - lam_s0
        rks = []
        if order == 2:
            rks.append((self._lam(sig_prev) - lam_s0) / h)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, torch.stack(R), torch.tensor(b), h_phi_1, B_h

    def step(self, model_output, timestep, sample, return_dict=True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:                                      # index_for_timestep (:630-637)
            tv = int(timestep) if not torch.is_tensor(timestep) else int(timestep.item())
            idx = (self.timesteps.detach().cpu() == tv).nonzero()      # the CURRENT table: video-to-video cuts it short (any2video.py:1029-1033)
            self._step_index = idx[1 if len(idx) > 1 else 0].item()
        i = self._step_index
        sig = self.sigmas
        model_output = model_output.to(torch.float32).contiguous()
        sample = sample.contiguous()
        m_t = ops.lincomb([sample, model_output], [1.0, -float(sig[i])])           # x0 = x - sigma*v  (:313-315)
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        if use_corrector:                                                 # UniC (:482-626)
            order = self.this_order
            m0 = self.model_outputs[-1]
            sig_t, sig_s0 = sig[i], sig[i - 1]
            rks, R, b, h_phi_1, B_h = self._bh(sig_t, sig_s0, order, sig[i - 2] if order == 2 else None)
            alpha_t = 1 - sig_t
            c_last = sig_t / sig_s0
            if order == 1:
                rho_last = torch.tensor(0.5)
                c_m1 = None
                c_m0 = -alpha_t * h_phi_1 + alpha_t * B_h * rho_last
            else:
                rhos_c = torch.linalg.solve(R, b)
                rho_last = rhos_c[-1]
                c_m1 = -alpha_t * B_h * rhos_c[0] / rks[0]
                c_m0 = -alpha_t * h_phi_1 + alpha_t * B_h * (rhos_c[0] / rks[0] + rho_last)
            c_mt = -alpha_t * B_h * rho_last
            ins, cs = [self.last_sample, m0, m_t], [float(c_last), float(c_m0), float(c_mt)]
            if c_m1 is not None:
                ins.append(self.model_outputs[-2]); cs.append(float(c_m1))
            sample = ops.lincomb(ins, cs)
        for j in range(self.solver_order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
        self.model_outputs[-1] = m_t
        if self.lower_order_final:
            this_order = min(self.solver_order, len(self.timesteps) - i)
        else:
            this_order = self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order                                           # UniP (:350-480)
        sig_t, sig_s0 = sig[i + 1], sig[i]
        rks, R, b, h_phi_1, B_h = self._bh(sig_t, sig_s0, order, sig[i - 1] if order == 2 else None)
        alpha_t = 1 - sig_t
        c_x = sig_t / sig_s0
        if order == 2:
            c_m1 = -alpha_t * B_h * 0.5 / rks[0]
            c_m0 = -alpha_t * h_phi_1 + alpha_t * B_h * 0.5 / rks[0]
            prev = ops.lincomb([sample, m_t, self.model_outputs[-2]], [float(c_x), float(c_m0), float(c_m1)])
        else:
            prev = ops.lincomb([sample, m_t], [float(c_x), float(-alpha_t * h_phi_1)])
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)

    def scale_model_input(self, sample, *a, **k):
        return sample


class EulerScheduler:
    """shared/utils/euler_scheduler.py:26-87."""
    is_stateful = False

    def __init__(self, num_train_timesteps=1000, use_timestep_transform=True):
        self.num_train_timesteps = num_train_timesteps
        self.use_timestep_transform = use_timestep_transform
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None, shift=5.0):
        self.num_inference_steps = num_inference_steps
        ts = list(np.linspace(self.num_train_timesteps, 1, num_inference_steps, dtype=np.float32)) + [0.0]
        ts = [torch.tensor([t]) for t in ts]
        if self.use_timestep_transform:
            out = []
            for t in ts:
                t = t / self.num_train_timesteps
                out.append(shift * t / (1 + (shift - 1) * t) * self.num_train_timesteps)
            ts = out[:-1]
        self.timesteps = torch.tensor(ts)
        return self.timesteps

    def step(self, model_output, timestep, sample, return_dict=True, **kwargs):
        if self.timesteps is None:
            raise ValueError("Timesteps are not set. Call set_timesteps first.")
        t_val = float(timestep.flatten()[0].item()) if torch.is_tensor(timestep) else float(timestep)
        idx = int(torch.argmin((self.timesteps - t_val).abs()).item())
        dt_raw = self.timesteps[idx] - self.timesteps[idx + 1] if idx + 1 < len(self.timesteps) else self.timesteps[idx]
        dt = dt_raw.item() / self.num_train_timesteps
        prev = ops.lincomb([sample.contiguous(), model_output.to(torch.float32).contiguous()], [1.0, -dt])
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)

    def scale_model_input(self, sample, *a, **k):
        return sample


class HipScheduler:
    """The `wan_sched_*` object of the C ABI behind the reference's scheduler surface (`set_timesteps`, `.timesteps`,
    `.sigmas`, `step(...)[0]`, `scale_model_input`): kind "unipc" = FlowUniPCMultistepScheduler(shift=1,
    use_dynamic_shifting=False), kind "euler" = EulerScheduler(use_timestep_transform=True).  What `WanAny2VHIP.generate`
    steps with on a GPU; the Python classes above carry the same algebra for hosts without one (control-flow tests)."""
    order = 1
    is_stateful = True

    def __init__(self, kind="unipc", num_train_timesteps=1000):
        from ctypes import byref, c_void_p
        from . import lib as _L
        if kind not in ("unipc", "euler"):
            raise NotImplementedError(f"HipScheduler: kind {kind!r}")
        self.kind, self.num_train_timesteps = kind, num_train_timesteps
        self._L = _L
        h = c_void_p()
        _L.check(_L.load().wan_sched_create(byref(h), 0 if kind == "unipc" else 1, num_train_timesteps), "wan_sched_create")
        self._h = h
        self.timesteps = self.sigmas = self.num_inference_steps = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.load().wan_sched_destroy(h)

    def set_timesteps(self, num_inference_steps, device=None, shift=5.0, **unused):
        from ctypes import c_double, c_float
        ts = (c_double * num_inference_steps)()
        sg = (c_float * (num_inference_steps + 1))()
        self._L.check(self._L.load().wan_sched_set_timesteps(self._h, num_inference_steps, float(shift), ts, sg), "wan_sched_set_timesteps")
        self.num_inference_steps = num_inference_steps
        if self.kind == "unipc":
            self.timesteps = torch.tensor([int(v) for v in ts], dtype=torch.int64, device=device)
            self.sigmas = torch.tensor(list(sg), dtype=torch.float32)
        else:
            self.timesteps = torch.tensor(list(ts), dtype=torch.float32)
        return self.timesteps

    def step(self, model_output, timestep, sample, return_dict=True, **kwargs):
        if self.timesteps is None:
            raise ValueError("Timesteps are not set. Call set_timesteps first.")
        t = float(timestep.flatten()[0].item()) if torch.is_tensor(timestep) else float(timestep)
        v = model_output.to(torch.float32).contiguous()
        x = sample.contiguous()
        ops._req(v, torch.float32, "model_output"); ops._req(x, torch.float32, "sample")
        if v.shape != x.shape:
            raise ValueError(f"model_output {tuple(v.shape)} and sample {tuple(x.shape)} differ")
        prev = torch.empty_like(x)
        self._L.check(self._L.load().wan_sched_step(self._h, self._L.ptr(v), t, self._L.ptr(x), self._L.ptr(prev), x.numel(),
                                                    self._L.stream_ptr()), "wan_sched_step")
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)

    def scale_model_input(self, sample, *a, **k):
        return sample


def cfg_combine(cond, uncond, guide_scale):
    """noise_pred = uncond + g * (cond - uncond)   (any2video.py:1722)"""
    return ops.cfg_combine(cond, uncond, guide_scale)


# ---- the remaining samplers WanAny2V.generate() can build (any2video.py:513-545) ---------------------------
def get_sampling_sigmas(sampling_steps, shift):
    """fm_solvers.py:22-27."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """fm_solvers.py:29-72 (the `sigmas=` / plain forms generate() uses)."""
    if timesteps is not None:
        raise NotImplementedError("custom timestep schedules are not used on the Wan path")
    if sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


class FlowDPMSolverMultistepScheduler:
    """`dpm++`: shared/utils/fm_solvers.py FlowDPMSolverMultistepScheduler in the configuration generate() builds
    (order 2, dpmsolver++, midpoint, flow_prediction, lower_order_final, final sigma zero).  Scalar algebra on the
    host as the reference (:455-468, :529-557); each update is one fused wan_lincomb over the fp32 latents."""
    order = 1

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", shift=1.0,
                 use_dynamic_shifting=False, algorithm_type="dpmsolver++", solver_type="midpoint",
                 lower_order_final=True, euler_at_final=False, final_sigmas_type="zero", **unused):
        if (solver_order, prediction_type, algorithm_type, solver_type, final_sigmas_type, use_dynamic_shifting) != \
                (2, "flow_prediction", "dpmsolver++", "midpoint", "zero", False):
            raise NotImplementedError("only the configuration WanAny2V.generate() uses is implemented")
        self.num_train_timesteps, self.shift = num_train_timesteps, shift
        self.lower_order_final, self.euler_at_final = lower_order_final, euler_at_final
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self._step_index = None

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self._timesteps_host = self.timesteps.cpu()
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        self._step_index = None

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            tv = int(timestep) if not torch.is_tensor(timestep) else int(timestep.item())
            idx = (self.timesteps.detach().cpu() == tv).nonzero()      # the CURRENT table: video-to-video cuts it short (any2video.py:1029-1033)
            self._step_index = idx[1 if len(idx) > 1 else 0].item()
        i, sig, n = self._step_index, self.sigmas, len(self.timesteps)
        lower_order_final = i == n - 1                      # final_sigmas_type == "zero" (:745-748)
        model_output = model_output.to(torch.float32).contiguous()
        sample = sample.to(torch.float32).contiguous()
        m0 = ops.lincomb([sample, model_output], [1.0, -float(sig[i])])           # x0 = x - sigma*v (:385-386)
        self.model_outputs = [self.model_outputs[1], m0]
        sigma_t, sigma_s0 = sig[i + 1], sig[i]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        c_x = sigma_t / sigma_s0
        c_d0 = -(alpha_t * (torch.exp(-h) - 1.0))
        if self.lower_order_nums < 1 or lower_order_final:                          # first order (:455-468)
            prev = ops.lincomb([sample, m0], [float(c_x), float(c_d0)])
        else:                                                                       # second order midpoint (:529-553)
            m1 = self.model_outputs[-2]
            r0 = (self._lam(sigma_s0) - self._lam(sig[i - 1])) / h
            c_d1 = 0.5 * c_d0 * (1.0 / r0)                                          # on D1 = (1/r0)(m0 - m1)
            prev = ops.lincomb([sample, m0, m1], [float(c_x), float(c_d0 + c_d1), float(-c_d1)])
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else SchedulerOutput(prev)

    def scale_model_input(self, sample, *a, **k):
        return sample


class FlowMatchScheduler:
    """`causvid`: shared/utils/basic_flowmatch.py:8-54.  generate() overwrites `.timesteps` / `.sigmas` with its fixed
    table after construction (any2video.py:515-517); that works here the same way."""
    is_stateful = False

    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False):
        if inverse_timesteps or reverse_sigmas:
            raise NotImplementedError("training-only options")
        self.num_train_timesteps, self.shift = num_train_timesteps, shift
        self.sigma_max, self.sigma_min, self.extra_one_step = sigma_max, sigma_min, extra_one_step
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False):
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            sig = torch.linspace(start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            sig = torch.linspace(start, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * sig / (1 + (self.shift - 1) * sig)
        self.timesteps = self.sigmas * self.num_train_timesteps

    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        ts = self.timesteps.detach().float().cpu()
        sg = self.sigmas.detach().float().cpu()
        tv = float(timestep.flatten()[0].item()) if torch.is_tensor(timestep) else float(timestep)
        tid = int(torch.argmin((ts - tv).abs()).item())
        sigma = sg[tid]
        sigma_ = 0.0 if (to_final or tid + 1 >= len(ts)) else sg[tid + 1]
        coef = float(torch.as_tensor(sigma_, dtype=torch.float32) - sigma)
        return [ops.lincomb([sample.to(torch.float32).contiguous(), model_output.to(torch.float32).contiguous()], [1.0, coef])]


class LCMScheduler:
    """`lcm`: shared/utils/lcm_scheduler.py:11-99."""

    def __init__(self, num_train_timesteps=1000, num_inference_steps=4, shift=1.0):
        self.num_train_timesteps, self.num_inference_steps, self.shift = num_train_timesteps, num_inference_steps, shift
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None, shift=None, **kwargs):
        self.num_inference_steps = min(num_inference_steps, 8)
        shift = self.shift if shift is None else shift
        t = torch.linspace(0, 1, self.num_inference_steps + 1, dtype=torch.float32)
        sigma_min = 0.003 / 1.002
        sig = sigma_min + (1.0 - sigma_min) * (1 - t)
        self.sigmas = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = self.sigmas[:-1] * self.num_train_timesteps
        self._sig_host, self._ts_host = self.sigmas.clone(), self.timesteps.clone()
        if device is not None:
            self.timesteps = self.timesteps.to(device)
            self.sigmas = self.sigmas.to(device)
        self._step_index = None

    def step(self, model_output, timestep, sample, **kwargs):
        if self._step_index is None:
            self._sig_host, self._ts_host = self.sigmas.clone(), self.timesteps.clone()      # the current tables (a caller may have cut them short)
            tv = float(timestep.flatten()[0].item()) if torch.is_tensor(timestep) else float(timestep)
            idx = (self._ts_host == tv).nonzero()
            self._step_index = idx[0].item() if len(idx) > 0 else int(torch.argmin((self._ts_host - tv).abs()).item())
        i = self._step_index
        nxt = self._sig_host[i + 1] if i + 1 < len(self._sig_host) else torch.zeros(())
        coef = float(nxt - self._sig_host[i])
        self._step_index += 1
        return SchedulerOutput(ops.lincomb([sample.to(torch.float32).contiguous(),
                                            model_output.to(torch.float32).contiguous()], [1.0, coef]))
