import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


Q_CASES = ["gas512", "ice512", "ortho2000_unwrapped", "sparse_clusters", "subpop_rows",
           "subpop_points", "tiny_box", "nonperiodic_z", "cut_2p9_4p0"]
EXTRA_CASES = [c for c in Q_CASES if c != "cut_2p9_4p0"]


def case_inputs(g):
    pos = g["pos"]
    sub = g["sub"] if "sub" in g.files else pos
    return sub, pos, g["box"], float(g["low"]), float(g["high"])


@pytest.fixture(scope="session")
def oracle():
    import wol_oracle
    wol_oracle.lib()
    # OpenMP would otherwise start one thread per core it can see; the GPU box shows far more cores than its CPU share
    wol_oracle.set_threads(min(8, len(os.sched_getaffinity(0))))
    return wol_oracle
