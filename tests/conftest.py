import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

FREQS = np.array([.00570, .00293, .00430, .00212, .03260, .10464, .00154, .00057, .01001, .00416, .01554, .00618,
                  .02498, .00262, .00140, .01412, .05563, .71097])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared libraries are build products (git-ignored): build whichever is missing before the first test needs it
    (hipcc cross-compiles gfx950 without a GPU; __graft_entry__.build() does the same)."""
    from epilogos_amd import build
    if not build.LIB_PATH.exists():
        build.build_library(force=False, verbose=False)
    if not build.IO_LIB_PATH.exists():
        build.build_io_library(force=False, verbose=False)


def free_port():
    """A TCP port that is free right now, for torch.distributed.run rendezvous in the multi-process tests."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def load_golden(name):
    return dict(np.load(GOLDEN / name, allow_pickle=False))


@pytest.fixture(scope="session")
def golden_real():
    return load_golden("real_slice.npz")


@pytest.fixture(scope="session")
def golden_synth():
    return load_golden("synth833.npz")


@pytest.fixture(scope="session")
def golden_s3():
    return load_golden("s3_small.npz")


@pytest.fixture(scope="session")
def golden_pair():
    return load_golden("paired.npz")


@pytest.fixture(scope="session")
def golden_edge():
    return load_golden("edge.npz")


def synth_states(R, N, S=18, seed=0, uniform=False):
    """Synthetic 0-based state matrix with the chr1 state frequencies or uniform states."""
    rng = np.random.default_rng(seed)
    if uniform:
        return rng.integers(0, S, size=(R, N)).astype(np.int8)
    p = FREQS[:S] / FREQS[:S].sum() if S <= 18 else np.full(S, 1.0 / S)
    return rng.choice(S, size=(R, N), p=p).astype(np.int8)
