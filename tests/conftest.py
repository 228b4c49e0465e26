import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def weights0():
    """Synthetic full-size weights, seed 0, as torch CPU tensors keyed by prefixed state-dict name."""
    import torch
    from oracle import sva_oracle as O
    from streamvoiceanon_amd import specs

    torch.set_grad_enabled(False)
    return O.load_synth_weights(0, specs.all_specs(prompt_path=True))


@pytest.fixture(scope="session")
def weights1():
    import torch
    from oracle import sva_oracle as O
    from streamvoiceanon_amd import specs

    torch.set_grad_enabled(False)
    return O.load_synth_weights(1, specs.tokenizer_specs())


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name + ".npz"))
