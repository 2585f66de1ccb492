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


def band_file(data: bytes, band) -> bytes:
    """A standalone JPEG file for a band of a restart-marked image (zj_image_band): the parent's headers with
    the frame height replaced by the band's, the band's entropy-coded bytes, EOI.  What a rank that only
    receives its band would be decoding; lets the CPU oracle decode the band on its own."""
    import ctypes as C
    sof = data.index(b"\xff\xc0")
    sos = data.index(b"\xff\xda", sof)
    scan0 = sos + 2 + ((data[sos + 2] << 8) | data[sos + 3])
    head = bytearray(data[:scan0])
    head[sof + 5], head[sof + 6] = band.height >> 8, band.height & 255
    return bytes(head) + C.string_at(band.image.scan, band.image.scan_len) + b"\xff\xd9"
