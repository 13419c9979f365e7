import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def asset_root(tmp_path_factory):
    """Directory holding small generated OBJ/MTL/PPM assets (see raytracer_amd.assets)."""
    from raytracer_amd import assets

    root = str(tmp_path_factory.mktemp("assets"))
    assets.write_room(root, "room", detail=1)
    assets.write_blob(root, "blob", subdivisions=3)
    assets.write_tree(root, "tree", n_leaves=6000)
    return root


@pytest.fixture(scope="session")
def renderer():
    """One GPU context for the whole session (fails loudly without a GPU)."""
    from raytracer_amd.renderer import Renderer

    r = Renderer(0)
    yield r
    r.close()
