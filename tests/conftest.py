import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("OASR_TESTING_HOOKS", "1")  # the kernel-path setters of include/oasr_testing.h are inert without it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def tiny_case():
    """Seeded tiny/B=2 case shared by oracle and GPU parity tests (SURVEY.md §8d generator)."""
    import numpy as np
    import torch
    from oracle import mel_oracle as me
    from oracle import model_oracle as mo
    dims = mo.VARIANTS["tiny"]
    sd = mo.init_state_dict(dims, seed=0)
    pcm, ti, ty, tl = mo.synthetic_batch([0, 1])
    mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
    return dict(dims=dims, sd=sd, pcm=pcm, tokens=ti, targets=ty, text_len=tl, mel=mel)
