import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def lib_built():
    """lib_gpboost_amd.so must exist (built by __graft_entry__.build()); tests never build it implicitly on the GPU box."""
    from gpboost_amd.libpath import find_lib_path
    return find_lib_path()
