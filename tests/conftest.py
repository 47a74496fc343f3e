import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(REPO, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def char_golden():
    return golden("char_model")


@pytest.fixture(scope="session")
def orc_char(oracle, char_golden):
    g = char_golden
    return oracle.make_char(g["parent"], g["local_translation"], g["local_rotation"], g["joint_type"],
                            g["joint_axis"], g["dof_idx"], int(g["dof_size"]))
