import os
import sys

import numpy as np
import pytest

# torch bundles its own ROCm runtime (libamdhip64 / librccl) with the same sonames as /opt/rocm's, which
# libptk.so links: whichever is loaded first serves both.  Loading torch first keeps one consistent
# runtime in the process for the tests that use torch.distributed next to the kernels.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once
    lib = os.path.join(ROOT, "pbrpathtracer_amd", "libptk.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pbrpathtracer_amd", "csrc"), "-j4", "-s"])


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def scene_from_golden(z):
    return {k[6:]: z[k] for k in z.files if k.startswith("scene_")}


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle_binding as OB
    OB.build()
    return OB
