import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def usim():
    """the product package (directory name has hyphens, so it is imported through importlib); built on demand like
    __graft_entry__.build() does, so that a fresh checkout can run the suite directly"""
    lib = ROOT / "robotic-ultrasound-imaging_amd" / "lib" / "libusim.so"
    if not lib.exists():
        import subprocess
        subprocess.run(["make", "-s", "-C", str(ROOT / "robotic-ultrasound-imaging_amd" / "csrc")], check=True)
    return importlib.import_module("robotic-ultrasound-imaging_amd")


@pytest.fixture(scope="session")
def pins():
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "reference_pins.npz")
