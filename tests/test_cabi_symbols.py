"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/dvo_hip.h declares; without a GPU the entry points fail loudly (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_functions():
    text = (ROOT / "include" / "dvo_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef DVO_EXPERIMENTS.*?#endif", "", text, flags=re.S)      # experiment builds only: not in the product library
    return sorted(set(re.findall(r"\b(dvo_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = _declared_functions()
    for must in ("dvo_create", "dvo_destroy", "dvo_set_camera", "dvo_upload_frame", "dvo_normal_equations",
                 "dvo_weighted_r2_sum", "dvo_residuals_jacobian", "dvo_solve_batch", "dvo_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from dvo_amd import _cabi
    lib = ctypes.CDLL(str(_cabi.library_path()))
    for name in _declared_functions():
        assert hasattr(lib, name), "libdvo_hip.so does not export {}".format(name)
    # and the Python binding covers exactly the header
    assert sorted(_cabi.SIGNATURES) == _declared_functions()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import dvo_amd
    with pytest.raises(dvo_amd.DVOHipError) as e:
        dvo_amd.HipContext()
    assert e.value.code in (-2, -3)
    cam = dvo_amd.RGBDCameraModel(np.eye(3, dtype=np.float32), 1.0)
    with pytest.raises(ValueError):           # get_dvo wraps any constructor failure (core/__init__.py:32-38)
        dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1)
    with pytest.raises(ValueError):
        dvo_amd.get_dvo("kerl", cam, dvo_amd.Se3.identity(), levels=1)


def test_create_rejects_bad_sizes_before_touching_the_device():
    import dvo_amd
    for kw in (dict(max_w=0), dict(max_h=-1), dict(max_levels=0), dict(max_levels=9), dict(n_slots=1), dict(max_pairs=0),
               dict(max_w=8192, max_h=4096)):          # (h+1)*pitch >= 2^24: float32 tap offsets would be inexact
        with pytest.raises(dvo_amd.DVOHipError) as e:
            dvo_amd.HipContext(**kw)
        assert e.value.code == -1, kw


def test_product_package_does_not_import_the_oracle():
    pkg = ROOT / "dense-visual-odometry_amd"
    pat = re.compile(r"^\s*(import|from)\s+\S*oracle", re.M)
    for path in pkg.rglob("*.py"):
        assert not pat.search(path.read_text()), path
    for path in list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        assert not re.search(r"#include\s+[\"<][^\">]*oracle", path.read_text()), path


def test_header_is_plain_c(tmp_path):
    """include/dvo_hip.h must be consumable by a C compiler (the boundary is a C ABI, not a C++ one) and a C program
    must link against the shared library with nothing but the header."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    from dvo_amd import _cabi
    src = tmp_path / "use_abi.c"
    src.write_text(
        '#include "dvo_hip.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "    dvo_ctx* ctx = 0;\n"
        "    int rc = dvo_create(0, 64, 48, 2, 2, 1, &ctx);\n"
        '    printf("%d %s\\n", rc, dvo_last_error(0));\n'
        "    if (rc == DVO_OK) dvo_destroy(ctx);\n"
        "    return 0;\n}\n")
    exe = tmp_path / "use_abi"
    lib_dir = _cabi.library_path().parent
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
           "-L", str(lib_dir), "-l:libdvo_hip.so", "-Wl,-rpath," + str(lib_dir), "-Wl,--allow-shlib-undefined"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    code = int(run.stdout.split()[0])
    assert code in (0, -2, -3)          # created (GPU box) or "no HIP device" (build container) — never a crash


def test_committed_kernel_facts_are_of_the_library_that_was_built():
    """profiles/kernel_isa.json (instruction hash, static instruction counts of the level-0 reduction's record loop) must describe the
    kernel in the library that `make` just built - bench.py marks recorded counter readings stale otherwise - and the issue-rate probe
    (dvo_debug_valu_rate) must issue exactly the composition of that loop.  Refresh with `python tools/isa_count.py --write` and the
    DVO_PROBE_* constants in csrc/dvo_kernels.hip when the kernel changes."""
    import json
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import isa_count
    from dvo_amd import _cabi
    try:
        facts = isa_count.kernel_facts(_cabi.library_path())
    except RuntimeError as e:
        pytest.skip(str(e))
    committed = json.loads((ROOT / "profiles" / "kernel_isa.json").read_text())
    assert committed["sha256"] == facts["sha256"], "run `python tools/isa_count.py --write` and commit profiles/kernel_isa.json"
    assert committed["loop"]["by_class"] == facts["loop"]["by_class"]
    src = (ROOT / "dense-visual-odometry_amd" / "csrc" / "dvo_kernels.hip").read_text()
    probe = {k: int(re.search(r"#define DVO_PROBE_%s (\d+)" % k, src).group(1)) for k in ("FAST", "CONFLICT", "SLOW", "RCP")}
    bc = facts["loop"]["by_class"]
    assert (probe["FAST"], probe["CONFLICT"], probe["SLOW"], probe["RCP"]) == (bc["fast"], bc["fast_bank_conflict"], bc["slow"], bc["trans"])
    assert 120 < facts["loop"]["valu_per_point"] < 160 and facts["loop"]["lds"] == 0

