import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The native parts are built in-tree beforehand; build them if this is a fresh checkout."""
    import __graft_entry__ as g

    need = [os.path.join(ROOT, "jpeg-decoder_amd", "libzj.so"), os.path.join(ROOT, "oracle", "libzjoracle.so"),
            os.path.join(ROOT, "tests", "sim", "libzjsim.so"), os.path.join(ROOT, "jpeg-decoder_amd", "synth", "libzjsynth.so")]
    if not all(os.path.exists(p) for p in need):
        g.build()


def fixture_bytes(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


FIXTURES = ["porsche.jpg", "profile.jpg", "test-baseline.jpg", "out.jpg", "large.jpg"]
