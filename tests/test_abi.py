"""CPU: the C-ABI library loads and exports every symbol include/wanhip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    so = os.path.join(ROOT, "wan2gp_amd", "libwanhip.so")
    if not os.path.isfile(so):
        import __graft_entry__ as g
        g.build()
    return so


def header_symbols():
    src = open(os.path.join(ROOT, "include", "wanhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wan_[a-z0-9_]+)\s*\(", src)) - {"wan_poll_fn", "wan_gather_fn"})


def test_header_symbols_exported(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    exported = set(re.findall(r" T (wan_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, f"declared in wanhip.h but not exported: {missing}"


def test_ctypes_binding_covers_header(built):
    from wan2gp_amd import lib
    L = lib.load()
    for s in header_symbols():
        assert s in lib.SIGNATURES, f"{s} has no ctypes signature"
        assert hasattr(L, s)
    assert L.wan_version() == 5          # 2: wan_dit_args.t_frames, wan_attention_bounded, wan_gemm_fp8; 3: wan_dit_args grew (n_vace ...), wan_sched_* / wan_vae_* / wan_sp_*; 4: wan_dit_args.nag_* / context_batches, wan_nag_combine; 5: wan_dit_args.perturbation_layers / x_id


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from wan2gp_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.WanHipError):
        lib.load()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "wan2gp_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle|/oracle/", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), f"{f} reaches into oracle/ (test infrastructure only)"


def test_cpu_tensors_are_rejected(built):
    import torch
    from wan2gp_amd import ops, lib
    x = torch.zeros(4, 256, dtype=torch.bfloat16)
    with pytest.raises(lib.WanHipError):
        ops.ln_affine(x, x[0], x[0])
