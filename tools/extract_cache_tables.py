"""BUILD TOOL -- writes wan2gp_amd/data/skip_cache_tables.json: the per-model calibration tables of the
reference's step-skipping caches (TeaCache rescale polynomials, MagCache magnitude ratios), i.e. the literals assigned inside
`family_handler.set_cache_parameters` (models/wan/wan_handler.py:172-214).  They are model calibration DATA the plugin has to
hand to `skip_steps_cache` exactly as the built-in handler does (wgp.py:7202); they are read out of the reference's source with
`ast` -- by the order of the assignments in each branch -- instead of being retyped.
Run in the build container:   python tools/extract_cache_tables.py"""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "wan2gp_amd", "data", "skip_cache_tables.json")


def main():
    src = os.path.join(REF, "models", "wan", "wan_handler.py")
    tree = ast.parse(open(src).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "family_handler")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "set_cache_parameters")
    mag, tea = [], []
    for n in ast.walk(fn):
        if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) and n.targets[0].id in ("def_mag_ratios", "coefficients"):
            (mag if n.targets[0].id == "def_mag_ratios" else tea).append((n.lineno, [float(ast.literal_eval(e)) for e in n.value.elts]))
    mag.sort(); tea.sort()
    # source order of the branches (wan_handler.py:182-213)
    mag_names = ["t2v_two_experts", "i2v_2_2", "ti2v_5B_with_start_image_and_source_video", "ti2v_5B", "t2v_1.3B", "i2v_720p", "i2v_480p", "t2v_14B"]
    tea_names = ["i2v_720p", "i2v_480p", "t2v_1.3B", "t2v_14B"]
    assert len(mag) == len(mag_names) and len(tea) == len(tea_names), (len(mag), len(tea))
    out = {"source": "models/wan/wan_handler.py set_cache_parameters", "lines": [fn.lineno, fn.end_lineno],
           "mag_ratios": {k: v for k, (_, v) in zip(mag_names, mag)}, "tea_coefficients": {k: v for k, (_, v) in zip(tea_names, tea)}}
    json.dump(out, open(OUT, "w"), indent=0)
    print("wrote", OUT, {k: len(v) for k, v in out["mag_ratios"].items()}, {k: len(v) for k, v in out["tea_coefficients"].items()})


if __name__ == "__main__":
    main()
