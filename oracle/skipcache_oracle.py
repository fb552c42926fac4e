"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's step-skipping caches (TeaCache / MagCache) on top of
oracle/wan_oracle.py: `WanModel.compute_magcache_threshold` / `compute_teacache_threshold` (models/wan/modules/model.py:
1373-1482) and the skip logic of `WanModel.forward` (model.py:1914-2064) for the t2v path (one or two streams).
Pinned against tests/golden/skipcache_tiny.npz, which oracle/make_golden_skipcache.py records from the reference's own
WanModel with `.cache` set (decisions, thresholds and outputs of every step).  Only tests/ may import this module."""
import numpy as np
import torch

from oracle import wan_oracle as O


class Cache:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def update(self, d):
        self.__dict__.update(d)


def magcache_threshold(cache, start_step, timesteps, speed_factor):
    """model.py:1373-1430."""
    n = len(timesteps)
    r = np.array([1.0] * 2 + list(cache.def_mag_ratios))
    if len(r) != 2 * n:
        def interp(a):
            if n == 1:
                return np.array([a[-1]])
            return a[np.round(np.arange(n) * ((len(a) - 1) / (n - 1))).astype(int)]
        r = np.stack([interp(r[0::2]), interp(r[1::2])], axis=1).reshape(-1)
    cache.mag_ratios = r
    target = int(n / speed_factor)
    best_t, best_d = 0.01, 1000
    th = 0.01
    while th <= 0.6:
        nb, d = 0, 1000
        err, steps, ratio = 0, 0, 1.0
        for i in range(n):
            skip = False
            if i > start_step:
                ratio *= r[2 * i]; steps += 1; err += abs(1 - ratio)
                if err < th and steps <= cache.magcache_K:
                    skip = True
                else:
                    err, steps, ratio = 0, 0, 1.0
            if not skip:
                nb += 1
                d = abs(target - nb)
        if d < best_d:
            best_t, best_d = th, d
        elif d > best_d:
            break
        th += 0.01
    cache.magcache_thresh = best_t
    return best_t


def _rel(e, prev):
    return ((e - prev).abs().mean() / prev.abs().mean()).cpu().item()


def teacache_threshold(cache, start_step, timesteps, speed_factor, W, cfg, dtype=torch.bfloat16):
    """model.py:1432-1482."""
    es = [O.time_embed(torch.stack([t]), W, cfg, dtype)[0] for t in timesteps]
    f = np.poly1d(cache.coefficients)
    n = len(es)
    target = int(n / speed_factor)
    best_t, best_d = 0.01, 1000
    th = 0.01
    while th <= 0.6:
        acc, nb, d = 0, 0, 1000
        for i in range(n):
            skip = False
            if not (i <= start_step or i == n - 1):
                acc += abs(f(_rel(es[i], es[i - 1])))
                if acc < th:
                    skip = True
                else:
                    acc = 0
            if not skip:
                nb += 1
                d = abs(target - nb)
        if d < best_d:
            best_t, best_d = th, d
        elif d > best_d:
            break
        th += 0.01
    cache.rel_l1_thresh = best_t
    return best_t


def decide(cache, n, x_id, step, e):
    """model.py:1914-1963."""
    joint = n > 1
    if cache.cache_type == "mag":
        if step <= cache.start_step:
            return [True] * n
        if cache.one_for_all and x_id != 0:
            return [cache.should_calc] * n
        flags = []
        for i in range(1 if cache.one_for_all else n):
            c = i if joint else x_id
            cache.accumulated_ratio[c] *= cache.mag_ratios[2 * step + c]
            cache.accumulated_steps[c] += 1
            cache.accumulated_err[c] += abs(1 - cache.accumulated_ratio[c])
            if cache.accumulated_err[c] < cache.magcache_thresh and cache.accumulated_steps[c] <= cache.magcache_K:
                flags.append(False)
                if i == 0 and x_id == 0:
                    cache.skipped_steps += 1
            else:
                flags.append(True)
                cache.accumulated_err[c], cache.accumulated_steps[c], cache.accumulated_ratio[c] = 0, 0, 1.0
        if cache.one_for_all:
            cache.should_calc = flags[0]
            return [flags[0]] * n
        return flags
    if x_id != 0:
        return [cache.should_calc] * n
    if step <= cache.start_step or step == cache.num_steps - 1 or cache.previous_modulated_input is None:
        calc = True
        cache.accumulated_rel_l1_distance = 0
    else:
        cache.accumulated_rel_l1_distance += abs(np.poly1d(cache.coefficients)(_rel(e, cache.previous_modulated_input)))
        if cache.accumulated_rel_l1_distance < cache.rel_l1_thresh:
            calc = False
            cache.skipped_steps += 1
        else:
            calc = True
            cache.accumulated_rel_l1_distance = 0
    cache.previous_modulated_input = e
    cache.should_calc = calc
    return [calc] * n


def dit_forward_cached(x_list, t, context_list, W, cfg, cache, x_id=0, real_step_no=0, dtype=torch.bfloat16, vace_context=None,
                       vace_scale=1.0):
    """WanModel.forward with `self.cache` set (model.py:1914-2064), t2v path (+ VACE context blocks, which a skipped stream skips
    together with its main blocks).  Returns (outputs, x_should_calc)."""
    hs = []
    grid = None
    for x in x_list:
        h, grid = O.patch_embed(x, W, cfg, dtype)
        hs.append(h)
    cos, sin = O.rope_tables(grid)
    e, e0 = O.time_embed(t, W, cfg, dtype)
    flags = decide(cache, len(hs), x_id, real_step_no, e)
    if cache.previous_residual is None:
        cache.previous_residual = [None] * len(hs)
    slots = list(range(len(hs))) if len(hs) > 1 else [x_id]
    for s, (sl, calc) in enumerate(zip(slots, flags)):
        if not calc:
            hs[s] = hs[s] + cache.previous_residual[sl]                      # x += previous_residual (:1967-1971)
    ori = [h.clone() if c else None for h, c in zip(hs, flags)]
    ctxs = [O.text_embed(c.to(dtype), W) for c in context_list]
    hints, scales = O.vace_hints(vace_context, vace_scale, W, cfg, len(hs))
    for i in range(cfg.num_layers):
        for s in range(len(hs)):
            if flags[s]:
                hs[s] = O.block_with_hints(hs[s], None if hints is None else hints[s], scales, e0, ctxs[s], cos, sin, W, i, cfg, False)
    for s, (sl, calc) in enumerate(zip(slots, flags)):
        if calc:
            cache.previous_residual[sl] = hs[s] - ori[s]                      # torch.sub(x, ori) (:2044-2062)
    outs = [O.unpatchify(O.head_forward(h, e, W, cfg), grid, cfg).float() for h in hs]
    return outs, flags
