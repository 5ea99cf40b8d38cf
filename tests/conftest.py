"""Shared fixtures. GPU tests are marked `gpu`; everything else must pass on a CPU-only box.

The oracle (oracle/) is the checker here and only here (plus smoke() and bench.py's cpu_baseline leg).
"""
import os
import sys

import numpy as np
import pytest

# The PyTorch ROCm wheel ships its own libamdhip64; libwfpt.so links the system one. Whichever is loaded first serves
# both (same SONAME), and only "torch first" works: with libwfpt first, torch later finds "No HIP GPUs" (seen on the
# GPU box when tests/test_gpu_parity.py ran on its own). bench.py imports torch first for the same reason. Tests that
# hand device pointers to torch (tiles.assemble_torch, the gloo tests) need it; the product itself never imports torch.
try:
    import torch  # noqa: F401  (must precede the first load of libwfpt.so in this process)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The oracle's ctypes binding (built on demand with its committed Makefile)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def wf():
    """The product package with libwfpt.so built (hipcc cross-compiles without a GPU)."""
    import wavefront_path_tracer_amd as W
    if not os.path.exists(W._build.LIB_PATH):
        W.build()
    W.lib()
    return W


@pytest.fixture(scope="session")
def gpu(wf):
    if wf.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests need an MI355X (there is no CPU fallback)")
    return wf


def bits(a):
    """View float arrays as integers so comparisons are bit-exact (and NaN-safe)."""
    a = np.ascontiguousarray(a)
    if a.dtype.fields:
        return a.view(np.uint8)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: shape/dtype {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba.reshape(ba.shape[0], -1).any(axis=1) if False else (ba != bb).reshape(ba.shape[0], -1).any(axis=1))
        first = int(bad[0][0])
        raise AssertionError(f"{what}: {len(bad)} of {ba.shape[0]} rows differ; first at {first}: {a[first]} vs {b[first]}")
