"""pytest configuration: registers the `gpu` marker and puts the package / oracle on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no compute calls).
`-m gpu`     : parity tests proper — HIP path (through the C-ABI) vs oracle / golden vectors.
Nothing here reads /root/reference (it does not exist on the GPU box).
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG_DIR = ROOT / "dense-visual-odometry_amd"
for p in (str(PKG_DIR), str(ROOT / "oracle"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def frames():
    """The reference's own fixture excerpt (tests/test_data: first ten frames of TUM fr1/xyz) as
    gray u8 / depth u16 arrays + intrinsics, committed as tests/golden/frames.npz."""
    f = np.load(GOLDEN / "frames.npz")
    return dict(gray=f["gray"], depth=f["depth"], K=f["intrinsics"], depth_scale=float(f["depth_scale"]),
                gt_T=f["gt_T"], level_intrinsics=f["level_intrinsics"], color_bgr_1=f["color_bgr_1"])
