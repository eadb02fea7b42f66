import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(GOLDEN, "cornell", "CornellBox-Original.mesh")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def cornell():
    from mipt import host
    return host.HostScene.load_obj(CORNELL, flip_winding=True)


@pytest.fixture(scope="session")
def cornell_oracle(cornell):
    import oracle_binding as ob
    return ob.OracleScene(cornell.desc)


@pytest.fixture(scope="session")
def atrium_small():
    """10k-triangle procedural atrium with 64x64 textures and normal maps."""
    from mipt import abi, host
    return host.HostScene.atrium(abi.SCENE_CRYTEK, 10000, 64, 1)


@pytest.fixture(scope="session")
def atrium_small_oracle(atrium_small):
    import oracle_binding as ob
    return ob.OracleScene(atrium_small.desc)


@pytest.fixture(scope="session")
def gpu():
    """Initialises the HIP library; GPU tests fail loudly if it is missing."""
    from mipt import device
    device.init(0)
    return device
