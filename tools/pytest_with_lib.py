#!/usr/bin/env python
"""Run pytest with another build of the library loaded as THE library (an experiment build beside libwanhip.so):
    python tools/pytest_with_lib.py libwanhip_conv.so tests/test_gpu_vae.py -q -m gpu
The file name is relative to wan2gp_amd/.  Not used by the product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import lib  # noqa: E402

lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
