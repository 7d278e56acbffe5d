import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Every parity test records what it MEASURED (not only whether it passed); the table is printed at the end of the run and
# appended to gpurun_out/parity_measured.txt, so the margin to each stated tolerance is on record.
PARITY = []


def record_parity(test, **measured):
    PARITY.append((test, measured))


def pytest_terminal_summary(terminalreporter):
    if not PARITY:
        return
    lines = [f"{t:58s} " + "  ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in m.items()) for t, m in PARITY]
    terminalreporter.write_sep("-", "parity: measured values")
    for ln in lines:
        terminalreporter.write_line(ln)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_measured.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import rgk_oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def product_lib():
    """The HIP product library (built in-tree).  No CPU fallback exists."""
    from rgk_amd import build, capi
    build.build(verbose=False)
    return capi.load_product()


@pytest.fixture(scope="session")
def cornell():
    from rgk_amd.workloads import Workload
    return Workload("cornell-256", scale=0.25, spp=8)


@pytest.fixture(scope="session")
def sponza_small():
    from rgk_amd.workloads import Workload
    return Workload("sponza-1080p", scale=0.1, spp=8)


def make_rays(o, d, near=0.0, far=10000.0):
    import numpy as np
    n = len(o)
    return np.concatenate([o, d, np.full((n, 1), near, np.float32), np.full((n, 1), far, np.float32)], axis=1).astype(np.float32)
