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
def golden():
    import util
    return util.Golden()


def require_ref(path, what=None):
    """The binaries under oracle/_ref (the unmodified reference and the drop-in, built where /root/reference exists)
    travel to the GPU box with the snapshot. Where a GPU is present their absence is a FAILURE -- the strongest parity
    tests would otherwise vanish without a trace; elsewhere (a clean clone on a CPU box) the test is skipped.
    HVK_REQUIRE_REF=1 / 0 forces either."""
    if os.path.exists(path):
        return
    want = os.environ.get("HVK_REQUIRE_REF")
    must = want == "1" or (want is None and os.path.exists("/dev/kfd"))
    msg = "%s not built (needs /root/reference at build time: python -c 'import __graft_entry__ as g; g.build()')" % (what or path)
    if must:
        pytest.fail(msg)
    pytest.skip(msg)
