"""Scratch: the GPU parity tests against another build of the library (e.g. one compiled with an experiment's
macro): python tools/exp/run_tests_with_lib.py <path of the .so> [pytest arguments].  Used to run every block through the
write pass's fall-back DC lookup (a build whose one-byte DC tables are all zero)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jpeg_decoder_amd import binding as B  # noqa: E402

B.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:] or [os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "--timeout", "200", "--timeout-method", "thread"]))
