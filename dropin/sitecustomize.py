"""Optional zero-argument form of `python -m nmrf_amd.dropin`: with this directory and the repository root on PYTHONPATH,

    PYTHONPATH=/path/to/repo/dropin:/path/to/repo python /path/to/NMRF/inference.py ...

Python imports `sitecustomize` at start-up, which installs the import hook of nmrf_amd/dropin.py before the reference
driver's first import; the driver and its command line stay unchanged.  A sitecustomize further down sys.path (the
distribution's) is still executed afterwards."""
import importlib.machinery
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
try:
    from nmrf_amd.dropin import install
    install()
except ImportError as e:                                   # repository root not on PYTHONPATH: say so, do not break start-up
    sys.stderr.write("[nmrf_amd dropin] not installed: %s\n" % e)
_spec = importlib.machinery.PathFinder.find_spec(
    "sitecustomize", [p for p in sys.path if p and os.path.abspath(p) != _here])
if _spec is not None and _spec.loader is not None:
    _mod = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_mod)
