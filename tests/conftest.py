import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A realigner context on cuda:0.  Fails loudly (no skip, no fallback) when the HIP library or the
    device is missing: `-m gpu` tests must exercise the native path."""
    from nanopore_amd import realign
    ctx = realign.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(autouse=True)
def _reset_context_options(request):
    """The test / bring-up switches a test set on the session's context (npr_ctx_option, include/nprealign.h) do not outlive it."""
    yield
    if "gpu_ctx" in request.fixturenames:
        from nanopore_amd import _lib
        ctx = request.getfixturevalue("gpu_ctx")
        for opt in _lib.OPTIONS.values():
            ctx.set_option(opt, 0)
