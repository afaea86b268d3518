import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def synth(seed, n, d, k=6, spread=4.0):
    """Same seeded Gaussian mixture as tests/golden/make_golden.py."""
    rs = np.random.RandomState(seed)
    centres = (spread * rs.standard_normal((k, d))).astype(np.float32)
    which = rs.randint(0, k, size=n)
    x = centres[which] + rs.standard_normal((n, d)).astype(np.float32)
    return x.astype(np.float32), which.astype(np.int32) + 1


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import RefHarness, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return RefHarness()


@pytest.fixture(scope="session")
def exdata():
    from som_lvq_pak_amd import textio
    out = {}
    tab = textio.LabelTable()
    for name in ("ex", "ex_fts"):
        out[name], _ = textio.read_entries(os.path.join(GOLDEN, "data", name + ".dat"))
    for name in ("ex1", "ex2"):
        out[name], _ = textio.read_entries(os.path.join(GOLDEN, "data", name + ".dat"), tab)
    out["lvq_init"], _ = textio.read_entries(os.path.join(GOLDEN, "cli", "lvq_init.cod"), tab)
    out["labels"] = tab
    return out


def load_trace(name):
    return np.load(os.path.join(GOLDEN, "traces", name + ".npz"))


def read_cod(name, table=None):
    from som_lvq_pak_amd import textio
    return textio.read_entries(os.path.join(GOLDEN, "cli", name), table)[0]
