"""pytest configuration: `gpu` marker + import paths for the oracle (test infrastructure)
and the product package (directory name `liquid-usrp_amd`, imported as liquid_usrp_amd)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_product():
    """Import liquid-usrp_amd/ (hyphenated directory) under the module name liquid_usrp_amd."""
    if "liquid_usrp_amd" in sys.modules:
        return sys.modules["liquid_usrp_amd"]
    pkg = os.path.join(ROOT, "liquid-usrp_amd")
    spec = importlib.util.spec_from_file_location(
        "liquid_usrp_amd", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["liquid_usrp_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def product():
    return load_product()
