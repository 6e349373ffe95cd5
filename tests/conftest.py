import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries (they are git-ignored): build them once, exactly as build() does
    lib = os.path.join(ROOT, "funscript_flow_amd", "csrc", "libffl_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
