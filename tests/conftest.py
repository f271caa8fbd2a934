import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


def load_package():
    """The product's Python face (directory name is not an identifier)."""
    return importlib.import_module("project2-pathtracer_amd")


@pytest.fixture(scope="session")
def pt():
    pkg = load_package()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    pkg.lib()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.lib()
    return orc


def has_reference():
    return os.path.exists("/root/reference/src/scene.cpp")
