import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", message=".*reduction: 'mean' divides.*")
warnings.filterwarnings("ignore", message=".*Sparse CSR tensor support is in beta.*")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_criterion():
    return np.load(os.path.join(GOLDEN, "criterion.npz"))


@pytest.fixture(scope="session")
def golden_ppi_teacher():
    return np.load(os.path.join(GOLDEN, "ppi_teacher.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_train():
    return np.load(os.path.join(GOLDEN, "train_arxiv.npz"), allow_pickle=False)


def as_t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)
