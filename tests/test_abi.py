"""The C-ABI library loads and exports every symbol include/tmx.h declares (no compute calls without a GPU),
and the ctypes mirror agrees with the C struct sizes."""
import ctypes as C
import os
import re

import pytest

from conftest import PRODUCT_LIB, ROOT
from trajopt_amd import abi, runtime


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tmx.h")).read()
    return sorted(set(re.findall(r"TMX_API\s+[\w\s\*]+?\b(tmx_\w+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(runtime.ABI_SYMBOLS)


@pytest.mark.parametrize("libname", ["product", "hostemu"])
def test_library_exports_every_declared_symbol(libname, hostemu_lib):
    path = PRODUCT_LIB if libname == "product" else hostemu_lib
    assert os.path.exists(path), f"{path} missing — run __graft_entry__.build()"
    lib = C.CDLL(path)
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"{sym} not exported by {path}"


def test_struct_sizes_match_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "tmx.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(tmx_joint),sizeof(tmx_link_sphere),sizeof(tmx_obstacle_sphere),sizeof(tmx_term),'
                   'sizeof(tmx_problem_desc),sizeof(tmx_sqp_params),sizeof(tmx_osqp_settings),sizeof(tmx_qp_record));return 0;}\n')
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    mirror = [C.sizeof(t) for t in (abi.Joint, abi.LinkSphere, abi.ObstacleSphere, abi.Term, abi.ProblemDesc,
                                    abi.SqpParams, abi.OsqpSettings, abi.QpRecord)]
    assert sizes == mirror


def test_runtime_refuses_without_extension(tmp_path):
    with pytest.raises(runtime.TmxError):
        runtime.load_library(str(tmp_path / "nope.so"))


def test_product_has_no_cpu_fallback():
    """without a HIP device tmx_create must fail loudly (status != OK), never fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tier")
    lib = runtime.load_library(PRODUCT_LIB)
    h = C.c_void_p()
    assert lib.tmx_create(0, C.byref(h)) != abi.TMX_OK
    with pytest.raises(runtime.TmxError):
        runtime.Context(0)
