import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the package directory `quadruped-robot_amd` under the module name quadruped_robot_amd."""
    if "quadruped_robot_amd" in sys.modules:
        return sys.modules["quadruped_robot_amd"]
    d = os.path.join(ROOT, "quadruped-robot_amd")
    spec = importlib.util.spec_from_file_location("quadruped_robot_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["quadruped_robot_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def ref(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref (compiled reference solvers) not available")
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """A live qrgpu context.  Fails (does not skip) when the HIP library is missing: the GPU
    tests must never pass on a fallback."""
    pkg._build.build()
    ctx = pkg.Context(device_id=0, max_batch=4096, horizon_max=16)
    yield ctx
    ctx.close()
