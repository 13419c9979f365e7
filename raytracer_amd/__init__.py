"""MI355X-native render loop for the jullanggit/raytracer scene format.

`parse` / `Scene` mirror the reference's `config::parse` / `Scene::render`;
`Renderer` is the thin wrapper over the C ABI of include/rt_mi355.h.
"""
from . import _ffi
from .scene import Scene, make_params, parse

__all__ = ["Scene", "make_params", "parse", "Renderer", "_ffi"]


def __getattr__(name):
    if name == "Renderer":
        from .renderer import Renderer
        return Renderer
    raise AttributeError(name)
