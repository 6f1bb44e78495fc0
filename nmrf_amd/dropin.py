"""Run the reference's own drivers (inference.py, main.py --eval-only) on the MI355X build, unchanged.

    python -m nmrf_amd.dropin /path/to/NMRF/inference.py --input L.png R.png --output out SOLVER.RESUME kitti.pth

The reference checkout stays where it is; nothing of it is copied.  `python script.py` always puts the script's own
directory FIRST on sys.path, so a shadowing directory on PYTHONPATH can never win against the checkout's `nmrf/` and
`ops/` packages.  Instead an import hook (a meta-path finder, consulted before the path-based finders) redirects exactly
the modules of the hot path and lets every other import fall through to the checkout:

    nmrf.models (+ .NMRF .DPN .NMP .backbone)      -> nmrf_amd.models.*       (build_model, NMRF; nmrf/models/__init__.py:9-10)
    ops, ops.functions, ops.modules                -> nmrf_amd.ops.*          (ops/functions/__init__.py, ops/modules/__init__.py)
    MultiScaleDeformableAttention                  -> ms_deform_attn_forward/backward on libnmrf_hip.so (ops/src/vision.cpp:13-16)
    nmrf.utils.frame_utils                         -> the checkout's module, plus `downsample_disp` (called at
                                                      nmrf/utils/evaluation.py:366 but defined nowhere in the checkout)
    nmrf.config                                    -> the checkout's (yacs) config; nmrf_amd.config only if yacs is absent

`install()` is idempotent and may also be called from user code / a notebook before `import nmrf`.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import runpy
import sys
import types

# reference module name -> nmrf_amd module that takes its place
ALIASES = {
    "nmrf.models": "nmrf_amd.models",
    "nmrf.models.NMRF": "nmrf_amd.models.nmrf",
    "nmrf.models.DPN": "nmrf_amd.models.dpn",
    "nmrf.models.NMP": "nmrf_amd.models.nmp",
    "nmrf.models.backbone": "nmrf_amd.models.backbone",
    "ops": "nmrf_amd.ops",
    "ops.functions": "nmrf_amd.ops.functions",
    "ops.functions.ms_deform_attn_func": "nmrf_amd.ops.functions",
    "ops.modules": "nmrf_amd.ops.modules",
    "ops.modules.ms_deform_attn": "nmrf_amd.ops.modules",
}
EXTENSION = "MultiScaleDeformableAttention"
PATCHED = "nmrf.utils.frame_utils"


def _extension_module():
    """The module the reference's `import MultiScaleDeformableAttention as MSDA` (ms_deform_attn_func.py:11) expects."""
    from .ops import functions as f
    m = types.ModuleType(EXTENSION, "HIP stand-in for the reference's CUDA extension (ops/src/vision.cpp:13-16)")
    m.ms_deform_attn_forward = f.ms_deform_attn_forward
    m.ms_deform_attn_backward = f.ms_deform_attn_backward
    return m


def _have(name):
    if name in sys.modules:
        return True
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        if self.target is None:
            return _extension_module()
        mod = importlib.import_module(self.target)          # the very same module object under a second name
        self._identity = (mod.__spec__, getattr(mod, "__loader__", None), getattr(mod, "__package__", None))
        return mod

    def exec_module(self, module):
        # importlib's module_from_spec() has just stamped the ALIAS spec / loader / package on the real nmrf_amd module
        # (_init_module_attrs with override for __spec__): put its own identity back, so that its relative imports keep
        # resolving inside nmrf_amd (no ImportWarning about __package__ != __spec__.parent) and importlib.reload works
        ident = getattr(self, "_identity", None)
        if ident is not None:
            module.__spec__, module.__loader__, module.__package__ = ident


class _PatchLoader(importlib.abc.Loader):
    """Runs the checkout's own loader, then adds what the checkout calls but does not define."""

    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        if not hasattr(module, "downsample_disp"):
            from . import frame_utils as ours
            module.downsample_disp = ours.downsample_disp


class DropinFinder(importlib.abc.MetaPathFinder):
    def __init__(self):
        self._busy = False

    def find_spec(self, fullname, path=None, target=None):
        if fullname in ALIASES:
            is_pkg = fullname in ("nmrf.models", "ops", "ops.functions", "ops.modules")
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(ALIASES[fullname]), is_package=is_pkg)
        if fullname == EXTENSION:
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(None))
        if fullname == "nmrf.config" and not _have("yacs"):
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader("nmrf_amd.config"), is_package=True)
        if fullname == PATCHED and not self._busy:
            self._busy = True
            try:
                spec = importlib.machinery.PathFinder.find_spec(fullname, path)
            finally:
                self._busy = False
            if spec is not None and spec.loader is not None:
                spec.loader = _PatchLoader(spec.loader)
            return spec
        return None


def install():
    """Put the finder in front of sys.meta_path (once).  Returns the finder."""
    for f in sys.meta_path:
        if isinstance(f, DropinFinder):
            return f
    finder = DropinFinder()
    sys.meta_path.insert(0, finder)
    return finder


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, DropinFinder)]
    for name in list(ALIASES) + [EXTENSION, PATCHED]:
        sys.modules.pop(name, None)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit("nmrf_amd.dropin: no such script: %s" % argv[0])
    install()
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))             # what `python script.py` would have done
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
