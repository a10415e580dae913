"""CPU tests of the drop-in boundary: libprt_hip.so loads, exports every entry point include/prt.h
declares, and refuses to run without a GPU (no CPU fallback). No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "prt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(prt_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("prt_create", "prt_destroy", "prt_set_scene", "prt_build_accel", "prt_set_uniforms",
                 "prt_render_iteration", "prt_render", "prt_sync", "prt_read_accum", "prt_last_error",
                 "prt_set_scramble_seed", "prt_upload_scrambles"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pathtracer_amd import capi

    assert os.path.exists(capi.LIB_PATH), "build with __graft_entry__.build()"
    L = C.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), f"{name} declared in include/prt.h but not exported"
    assert sorted(capi.EXPORTS) == _declared()


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "prt.h"\nint main(void){return sizeof(prt_uniforms)==464 && sizeof(prt_vertex)==48 '
                   '&& sizeof(prt_material)==96 && sizeof(prt_bvh_node)==32 && sizeof(prt_ray)==32 ? 0 : 1;}\n')
    for cc, flags in (("gcc", ["-std=c99"]), ("g++", ["-std=c++17", "-x", "c++"])):
        exe = tmp_path / ("t_" + cc)
        subprocess.check_call([cc, *flags, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_no_gpu_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pathtracer_amd import capi

    with pytest.raises(capi.PrtError) as e:
        capi.Context(8, 8)
    assert "no HIP device" in str(e.value) or "code" in str(e.value)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under pathtracer_amd/ or include/ may reference it."""
    bad = []
    for base in ("pathtracer_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hpp", ".h", ".hip", ".cpp", "Makefile")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"liboracle|pyoracle|oracle/|from oracle|import oracle", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
