"""TEST INFRASTRUCTURE ONLY -- tests/golden/fp8_linear.npz from the REFERENCE's own scaled-fp8 code
(shared/qtypes/scaled_fp8.py), executed on CPU.  Run in the build container:   python oracle/make_golden_fp8.py [linear] [forward]

The module imports optimum.quanto (absent here), so its pieces are lifted with `ast`, bodies untouched:
  module level:  _FP8_RANGE, _reshape_scale, _normalize_scaled_mm_scale, _scaled_mm_weight_scale, _quantize_activation
  methods of ScaledFP8WeightTensor, run on a stand-in object with the attributes they read (_data, _scale, dtype, device):
                 dequantize, _linear_fallback, _linear_scaled
ONE edit is made to `_linear_scaled`: the sub-expression `input.device.type != "cuda"` of its eligibility test (:327-334) is
replaced by `False`, because the fixture is produced on CPU (torch._scaled_mm has a CPU kernel with the same contract: fp32
accumulation of exact fp8 products, scales, optional bias, one rounding to out_dtype).  Everything after that test -- the
activation quantisation, the scaled matmul, the per-row output scale and bias -- is the reference's statements.
"""
import ast
import os
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "shared/qtypes/scaled_fp8.py")
OUT = os.path.join(ROOT, "tests", "golden", "fp8_linear.npz")


class _NotCuda(ast.NodeTransformer):
    def visit_Compare(self, node):
        if ast.unparse(node) == "input.device.type != 'cuda'":
            return ast.copy_location(ast.Constant(False), node)
        return self.generic_visit(node)


def lift():
    tree = ast.parse(open(SRC).read())
    want = {"_reshape_scale", "_normalize_scaled_mm_scale", "_scaled_mm_weight_scale", "_quantize_activation", "_normalize_default_dtype"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    body += [n for n in tree.body if isinstance(n, ast.Assign) and any(getattr(t, "id", "") in ("_FP8_RANGE", "_SCALED_FP8_DEFAULT_DTYPE") for t in n.targets)]
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ScaledFP8WeightTensor")
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("dequantize", "_linear_fallback", "_linear_scaled")]
    for m in meths:
        m.decorator_list = []
        if m.name == "_linear_scaled":
            _NotCuda().visit(m)
    assert {n.name for n in body if isinstance(n, ast.FunctionDef)} == want and len(meths) == 3
    ns = {"torch": torch}
    mod = ast.Module(body=body + meths, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, SRC, "exec"), ns)
    return ns


def bits(t):
    return t.view(torch.uint8 if t.element_size() == 1 else torch.int16).numpy().copy()


def main():
    ns = lift()
    g = torch.Generator().manual_seed(88)
    out = {}
    f8 = torch.float8_e4m3fn
    cases = [("per_tensor_bias", (70, 256), 128, "tensor", True), ("per_row_bias", (70, 256), 128, "row", True),
             ("per_row_col_nobias", (33, 128), 64, "rowcol", False), ("batched_3d_per_row", (2, 40, 192), 96, "row", True),
             ("zero_input", (16, 64), 32, "row", True)]
    for name, xs, N, kind, has_bias in cases:
        K = xs[-1]
        x = (torch.randn(*xs, generator=g) * 1.7).to(torch.bfloat16)
        if name == "zero_input":
            x = torch.zeros_like(x)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        if kind == "tensor":
            scale = (w.abs().max() / 448).reshape(())
            wq = (w / scale).clamp(-448, 448).to(f8)
        else:
            scale = (w.abs().amax(dim=1) / 448)
            wq = (w / scale[:, None]).clamp(-448, 448).to(f8)
            if kind == "rowcol":
                scale = scale.reshape(N, 1)
        bias = (0.1 * torch.randn(N, generator=g)).to(torch.bfloat16) if has_bias else None
        me = types.SimpleNamespace(_data=wq, _scale=scale.float(), dtype=torch.bfloat16, device=wq.device)
        me.dequantize = lambda dtype=None, device=None, me=me: ns["dequantize"](me, dtype, device)
        xq, sa = ns["_quantize_activation"](x.reshape(-1, K), f8)
        o_scaled = ns["_linear_scaled"](me, x.clone(), None if bias is None else bias.clone())
        o_fb = ns["_linear_fallback"](me, x.clone(), None if bias is None else bias.clone())
        out[name + "/x"] = bits(x); out[name + "/w"] = bits(wq); out[name + "/scale"] = scale.float().numpy()
        if bias is not None:
            out[name + "/bias"] = bits(bias)
        out[name + "/x_fp8"] = bits(xq); out[name + "/scale_a"] = sa.numpy()
        out[name + "/out_scaled"] = bits(o_scaled); out[name + "/out_fallback"] = bits(o_fb)
        out[name + "/dequant"] = bits(ns["dequantize"](me))
        assert o_scaled.dtype == torch.bfloat16 and o_scaled.shape == (*xs[:-1], N)
        print(name, "scaled vs fallback rel diff", ((o_scaled.float() - o_fb.float()).norm() / o_fb.float().norm().clamp_min(1e-9)).item())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


def gen_forward_fp8():
    """tests/golden/forward_tiny_fp8.npz: the reference's WanModel (tiny config) whose block Linears run the reference's own
    `_linear_scaled` on fp8 weights -- each nn.Linear of blocks.* gets `forward = lambda x: _linear_scaled(stand-in, x, bias)`,
    the plan QLinearScaledFP8.forward takes on an fp8-capable GPU (:546-561).  Per-row and per-tensor weight scales."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import ref_shim, wan_oracle as O
    from oracle.make_golden import build_ref_model, ref_forward, f32
    ns_ref = ref_shim.load()
    ns = lift()
    cfg = O.make_config("tiny")
    f, h, w = 3, 8, 12
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([637], dtype=torch.int64)
    out = {"shape": np.array([f, h, w]), "t": np.array([637])}
    for tag, per_row in (("row", True), ("tensor", False)):
        Wb = O.synth_weights(cfg)
        W8 = O.quantize_checkpoint_fp8(Wb, per_row=per_row)
        m = build_ref_model(ns_ref, cfg, Wb, torch.bfloat16)
        n_patched = 0
        for name, mod in m.named_modules():
            key = name + ".weight"
            if isinstance(mod, torch.nn.Linear) and key in W8 and W8[key].dtype == torch.float8_e4m3fn:
                me = types.SimpleNamespace(_data=W8[key], _scale=W8[name + ".scale_weight"], dtype=torch.bfloat16, device=W8[key].device)
                me.dequantize = lambda dtype=None, device=None, me=me: ns["dequantize"](me, dtype, device)
                mod.forward = (lambda x, me=me, mod=mod: ns["_linear_scaled"](me, x, mod.bias))
                n_patched += 1
        assert n_patched == 10 * cfg.num_layers, n_patched
        r = ref_forward(ns_ref, m, [lat, lat], t, [ctx, ctx_null])
        out[f"cond_{tag}"], out[f"uncond_{tag}"] = f32(r[0]), f32(r[1])
        o = O.dit_forward([lat, lat], t, [ctx, ctx_null], W8, cfg, dtype=torch.bfloat16)
        print(f"forward_tiny_fp8[{tag}]: oracle bit-equal to the reference: {torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "forward_tiny_fp8.npz"), **out)


if __name__ == "__main__":
    import sys
    which = sys.argv[1:] or ["linear", "forward"]
    if "linear" in which:
        main()
    if "forward" in which:
        gen_forward_fp8()
