import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lbm():
    import mpilattice_boltzmann_amd as pkg
    pkg.build()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def digests():
    with open(os.path.join(GOLDEN, "digests.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def deck_paths(name, digests):
    d = digests[name]
    return os.path.join(GOLDEN, "decks", d["params"]), os.path.join(GOLDEN, "decks", d["obstacles"])
