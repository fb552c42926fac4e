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
    assert L.wan_version() == 8          # 8: wan_dit_args.context_key (the text cache), wan_vae_conv3d_ex, wan_gemm_debug_force16s; 7: wan_sp_info.a2a_chunks (chunked Ulysses q / o exchanges), wan_permute16_ex, wan_debug_delay; 6: wan_sp_info.mode / a2a_begin / a2a_wait (Ulysses), wan_sp_a2a_begin, wan_permute16; 2: wan_dit_args.t_frames, wan_attention_bounded, wan_gemm_fp8; 3: wan_dit_args grew (n_vace ...), wan_sched_* / wan_vae_* / wan_sp_*; 4: wan_dit_args.nag_* / context_batches, wan_nag_combine; 5: wan_dit_args.perturbation_layers / x_id


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


# ---- the ctypes table against the header's declarations: arity, argument classes, struct layouts -------------------------------
_FN_PTR_TYPES = {"wan_poll_fn", "wan_gather_begin_fn", "wan_gather_wait_fn"}


def _c_class(decl):
    """'const float* const* x' -> 'ptr'; 'int64_t n' -> 'i64'; works on a declaration with or without the name."""
    d = decl.strip()
    if "*" in d or any(re.search(r"\b%s\b" % t, d) for t in _FN_PTR_TYPES):
        return "ptr"
    for pat, cls in ((r"\bu?int64_t\b", "i64"), (r"\bdouble\b", "f64"), (r"\bfloat\b", "f32"), (r"\bint\b", "i32"), (r"\bvoid\b", "void")):
        if re.search(pat, d):
            return cls
    raise AssertionError(f"unclassified C declaration: {decl!r}")


def _ctypes_class(t):
    import ctypes
    if t is None:
        return "void"
    if t in (ctypes.c_int,):
        return "i32"
    if t in (ctypes.c_int64, ctypes.c_uint64):
        return "i64"
    if t is ctypes.c_float:
        return "f32"
    if t is ctypes.c_double:
        return "f64"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or isinstance(t, type(ctypes.CFUNCTYPE(None))):
        return "ptr"
    raise AssertionError(f"unclassified ctypes type: {t!r}")


def _parse_header():
    src = open(os.path.join(ROOT, "include", "wanhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            first, *rest = [s.strip() for s in stmt.split(",")]
            m = re.match(r"(.*?)(\w+)$", first)
            base, fname = m.group(1), m.group(2)
            fields.append((fname, _c_class(base + " x")))
            for r in rest:                                              # 'int F, H, W': the base type without its pointer stars
                stars = r.count("*")
                fields.append((r.replace("*", "").strip(), "ptr" if stars else _c_class(base.replace("*", "") + " x")))
        structs[name] = fields
    src_nostruct = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    src_nostruct = re.sub(r"typedef[^;]*;", "", src_nostruct)
    funcs = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(wan_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src_nostruct, flags=re.S):
        args = " ".join(args.split())
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        funcs[name] = (_c_class(ret + " r") if ret.strip() != "void" else "void", [_c_class(a) for a in arglist])
    return structs, funcs


def test_ctypes_signatures_match_the_header_argument_by_argument():
    """Names alone do not catch an argtype drift (wan_dit_args grew three times in one round): every prototype's return
    class, arity and argument classes (pointer / int / int64_t / float / double) must equal lib.SIGNATURES'."""
    from wan2gp_amd import lib
    _, funcs = _parse_header()
    assert set(funcs) == set(header_symbols()) and len(funcs) > 80
    for name, (ret, args) in sorted(funcs.items()):
        res, argtypes = lib.SIGNATURES[name]
        got = [_ctypes_class(a) for a in argtypes]
        assert len(got) == len(args), f"{name}: header has {len(args)} arguments, ctypes table {len(got)}"
        assert got == args, f"{name}: header {args} vs ctypes {got}"
        want_ret = "ptr" if name == "wan_last_error" else ret
        assert _ctypes_class(res) == want_ret, f"{name}: return {ret} vs ctypes {_ctypes_class(res)}"


def test_ctypes_structs_match_the_header_field_by_field():
    """wan_dit_args / wan_dit_config / wan_sp_info: field names, order, classes and the resulting size (natural alignment)."""
    import ctypes
    from wan2gp_amd import lib
    structs, _ = _parse_header()
    size_of = {"i32": 4, "f32": 4, "i64": 8, "f64": 8, "ptr": 8}
    for cname, cls in (("wan_dit_args", lib.DitArgs), ("wan_dit_config", lib.DitConfig), ("wan_sp_info", lib.SpInfo)):
        want = structs[cname]
        got = [(n, _ctypes_class(t)) for n, t in cls._fields_]
        assert got == want, f"{cname}: header {want} vs ctypes {got}"
        off = 0
        for _, c in want:
            a = size_of[c]
            off = (off + a - 1) // a * a + a
        size = (off + 7) // 8 * 8 if any(size_of[c] == 8 for _, c in want) else off
        assert ctypes.sizeof(cls) == size, (cname, ctypes.sizeof(cls), size)
