"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`  : oracle vs. committed golden vectors, host logic, C-ABI symbol/loading checks (CPU only).
`-m gpu`        : parity tests proper -- HIP kernels (through the C ABI) vs. the oracle and the goldens.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
# accounting of the digest tests' repeat-once policy (tests/test_gpu_t3_fixture.py): every 200-iteration comparison with a reference-made
# digest that ran, and every one that needed its second try -- printed at the end of the session, also under -q
DIGEST_RUNS: list = []
SECOND_TRY: list = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if DIGEST_RUNS:
        terminalreporter.write_line(f"[library-flake accounting] {len(SECOND_TRY)} of {len(DIGEST_RUNS)} digest comparisons needed a second run "
                                    f"(each of those tests ends as XFAIL, not as a pass)")
        for note in SECOND_TRY:
            terminalreporter.write_line("[library-flake accounting]   " + note)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU: skip them unless CUDA(HIP) is visible."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_library_built():
    """A fresh checkout has no .so (it is git-ignored): compile it once per session.  This is the native code itself,
    not a fallback -- every op still fails loudly if the library cannot be built or loaded."""
    from auto_round_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    yield


def pytest_runtest_setup(item):
    """The C oracle restates `tensor / python_scalar` either as torch evaluates it on the CPU (a division: the semantics of the
    CPU-generated golden vectors) or as torch's GPU kernels do (a multiplication by the reciprocal: what the reference computes when
    it runs on the MI355X, and what the HIP kernels follow).  GPU parity tests compare against the GPU semantics; everything else
    keeps the CPU semantics.  Only the asymmetric INT schemes can tell the two apart (oracle/ar_oracle.c div_py_scalar)."""
    try:
        from oracle import oracle as orc

        orc.set_scalar_div_mode("gpu" in item.keywords)
    except Exception:  # pragma: no cover  (oracle not built yet: the tests that need it build it)
        pass
