import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_tables():
    return load_golden("motion_tables.npz")


@pytest.fixture(scope="session")
def golden_motion_state():
    return load_golden("motion_state.npz")


@pytest.fixture(scope="session")
def golden_task_ops():
    return load_golden("task_ops.npz")


@pytest.fixture(scope="session")
def golden_env_trace():
    return load_golden("env_trace.npz")


# ---- the large reference-pinned fixtures (oracle/gen_golden_large.py: OUTPUTS of the reference at BASELINE config 2's size; the inputs are
# regenerated from their seeds by oracle/golden_inputs.py)
@pytest.fixture(scope="session")
def golden_task_ops_1024():
    return load_golden("task_ops_1024.npz")


@pytest.fixture(scope="session")
def golden_motion_state_4096():
    return load_golden("motion_state_4096.npz")


@pytest.fixture(scope="session")
def golden_env_trace_1024():
    return load_golden("env_trace_1024.npz")
