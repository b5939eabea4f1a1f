"""Make the reference's Python importable in THIS container (test infrastructure only).

`install()` puts /root/reference on sys.path and registers inert stand-ins for the
third-party modules that are absent here: isaacgym.gymapi/gymtorch/gymutil,
imageio, lxml, stl, vtk, mujoco_py, glfw, smplx, ...  The stand-ins are attribute sinks:
they let `import` statements succeed, they are never called on the paths we execute
(motion_lib, torch_utils, the @torch.jit.script reward/reset/obs functions, poselib FK).
Never used on the GPU box (/root/reference does not exist there).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = (
    "imageio", "lxml", "stl", "vtk", "mujoco_py", "glfw", "smplx", "joblib_stub",
    "cv2", "ipdb", "wandb", "gym", "rl_games", "horovod", "tensorboardX", "chumpy",
    "OpenGL", "pyvista", "open3d", "trimesh", "numpy_stl", "autograd", "matplotlib",
    "mpl_toolkits",
)
_STUB_EXACT = ("isaacgym.gymapi", "isaacgym.gymtorch", "isaacgym.gymutil")


class _Sink(types.ModuleType):
    """Module whose every attribute is another sink / a dummy class."""

    __path__ = []  # behaves as a package so that `import a.b.c` works

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            import torch.nn as nn
            cls = type(name, (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self)})
            setattr(self, name, cls)
            return cls
        sub = _Sink(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return None


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if fullname in _STUB_EXACT or root in _STUB_ROOTS:
            try:  # prefer a real install when one exists
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Sink(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present; goldens can only be regenerated where /root/reference exists")
    sys.dont_write_bytecode = True  # never leave __pycache__ in the read-only mount
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "poselib"), os.path.join(REFERENCE_ROOT, "embodied_pose")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.meta_path.append(_Finder())
    _installed = True
