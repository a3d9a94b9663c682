import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


PARITY_NOTES = []


def parity_note(msg):
    """one line for the "parity" section pytest prints after the run (also with -q): what was compared with the
    reference and whether it was bit-identical — so that the driver's GPUTEST record says so"""
    PARITY_NOTES.append(msg)


def pytest_terminal_summary(terminalreporter):
    if PARITY_NOTES:
        terminalreporter.section("parity vs the compiled reference")
        for line in PARITY_NOTES:
            terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import jpeg2png_amd
    jpeg2png_amd.build()
    return jpeg2png_amd.load_library()


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    bindings.oracle_lib()
    return bindings


def psnr(a, b):
    """per-plane PSNR, peak 255."""
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def bit_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                                                 np.ascontiguousarray(b, np.float32).view(np.uint32))


def make_case(W, H, sub, quality, seed, y_only=False):
    """synthetic planes with fdata decoded by the CPU oracle (jpeg.c:83-92 + unbox)."""
    from jpeg2png_amd import synth
    from oracle import bindings
    planes = synth.make_planes(W, H, sub, quality, seed=seed, y_only=y_only)
    for p in planes:
        p.fdata = bindings.decode_plane(p)
    return planes
