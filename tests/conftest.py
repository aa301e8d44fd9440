import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need an MI355X and the built extension: skip (not fail) where either is missing."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    reason = None
    if not have_gpu:
        reason = "no GPU visible (run through gpurun)"
    else:
        try:
            import diff_gaussian_rasterization  # noqa: F401
        except ImportError as exc:   # on a GPU box a missing extension must be LOUD, not a silent skip
            raise pytest.UsageError(f"GPU present but the HIP extension is not built: {exc}")
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture
def option():
    """option(name, value): set a product-library option for the duration of one test."""
    from util import set_option
    undo = []

    def _set(name, value):
        undo.append((name, set_option(name, value)))
    yield _set
    for name, old in reversed(undo):
        set_option(name, old)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle
