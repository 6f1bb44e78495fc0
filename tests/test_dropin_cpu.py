"""The drop-in boundary: the reference's drivers import `nmrf.*` / `ops.*` from THEIR checkout; with the
import hook of nmrf_amd/dropin.py the hot-path modules come from nmrf_amd and everything else still resolves in the
checkout.  The checkout here is a stub tree with the reference's layout (the reference itself never travels); the two
scripts carry the import lines of inference.py:7-12 and main.py:14-19."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INFERENCE_IMPORTS = """
from nmrf.config import get_cfg
from nmrf.utils.logger import setup_logger
from nmrf.data import datasets
from nmrf.utils import frame_utils
from nmrf.utils import visualization
from nmrf.models import build_model
"""
MAIN_IMPORTS = """
from nmrf.data import build_train_loader, build_val_loader
from nmrf.models import build_model
from nmrf.utils import misc
import nmrf.utils.dist_utils as comm
from nmrf.utils.logger import setup_logger
from nmrf.utils import evaluation
"""
REPORT = """
import sys, nmrf, nmrf.utils
import MultiScaleDeformableAttention as MSDA
from ops.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch
from ops.modules import MSDeformAttn
from nmrf.models.NMRF import NMRF
print("build_model", build_model.__module__)
print("NMRF", NMRF.__module__)
print("msda", MSDA.ms_deform_attn_forward.__module__, MSDeformAttnFunction.__module__, MSDeformAttn.__module__)
print("stub_pkg", nmrf.STUB, setup_logger.__module__)
print("argv", sys.argv[1:])
"""


def _stub_checkout(tmp):
    """nmrf/{config,data,models,utils} + ops/ with the reference's layout; nmrf.models and ops RAISE when executed."""
    def w(rel, body=""):
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(body))
    w("nmrf/__init__.py", "STUB = 'checkout'\n")
    w("nmrf/config/__init__.py", "def get_cfg():\n    return 'stub-cfg'\n")
    w("nmrf/data/__init__.py", "from .datasets import build_train_loader, build_val_loader\n")
    w("nmrf/data/datasets.py", "from nmrf.utils import frame_utils, misc, evaluation\nfrom nmrf.utils import dist_utils as comm\n"
                               "def build_train_loader(*a):\n    pass\ndef build_val_loader(*a):\n    pass\n")
    w("nmrf/utils/__init__.py")
    w("nmrf/utils/logger.py", "def setup_logger(*a, **k):\n    pass\ndef log_every_n_seconds(*a):\n    pass\n")
    w("nmrf/utils/frame_utils.py", "def writeDispKITTI(*a):\n    pass\ndef read_gen(*a):\n    pass\nclass InputPadder:\n    pass\n")
    w("nmrf/utils/visualization.py")
    w("nmrf/utils/misc.py", "import nmrf.utils.dist_utils as comm\n")
    w("nmrf/utils/dist_utils.py")
    w("nmrf/utils/evaluation.py", "from nmrf.utils.logger import log_every_n_seconds\nfrom nmrf.utils import frame_utils\n"
                                  "HAS_DS = hasattr(frame_utils, 'downsample_disp')\n")
    w("nmrf/models/__init__.py", "raise ImportError('the checkout nmrf.models must be shadowed by nmrf_amd')\n")
    w("ops/__init__.py", "raise ImportError('the checkout ops package (CUDA extension) must be shadowed')\n")
    w("inference.py", INFERENCE_IMPORTS + REPORT + "print('frame_utils', frame_utils.__file__, hasattr(frame_utils, 'writeDispKITTI'), "
                                                   "frame_utils.downsample_disp.__module__)\n")
    w("main.py", MAIN_IMPORTS + REPORT + "print('evaluation', evaluation.HAS_DS)\n")


def _run(cmd, env_extra, cwd):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=cwd, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())


def _check(out, tmp):
    assert out["build_model"] == "nmrf_amd.models"
    assert out["NMRF"] == "nmrf_amd.models.nmrf"
    assert out["msda"] == "nmrf_amd.ops.functions nmrf_amd.ops.functions nmrf_amd.ops.modules"
    assert out["stub_pkg"] == "checkout nmrf.utils.logger"          # everything else still comes from the checkout
    assert out["argv"] == "['--input', 'a.png']"


def test_launcher_runs_reference_drivers_with_hot_path_from_nmrf_amd(tmp_path):
    tmp = str(tmp_path)
    _stub_checkout(tmp)
    out = _run([sys.executable, "-m", "nmrf_amd.dropin", os.path.join(tmp, "inference.py"), "--input", "a.png"],
               {"PYTHONPATH": ROOT}, cwd="/")
    _check(out, tmp)
    assert out["frame_utils"] == "%s True nmrf_amd.frame_utils" % os.path.join(tmp, "nmrf", "utils", "frame_utils.py")
    out = _run([sys.executable, "-m", "nmrf_amd.dropin", os.path.join(tmp, "main.py"), "--input", "a.png"],
               {"PYTHONPATH": ROOT}, cwd="/")
    _check(out, tmp)
    assert out["evaluation"] == "True"                              # evaluation.py:366 finds downsample_disp


def test_sitecustomize_form_keeps_the_command_line_unchanged(tmp_path):
    tmp = str(tmp_path)
    _stub_checkout(tmp)
    out = _run([sys.executable, os.path.join(tmp, "inference.py"), "--input", "a.png"],
               {"PYTHONPATH": os.path.join(ROOT, "dropin") + os.pathsep + ROOT}, cwd="/")
    _check(out, tmp)


def test_without_the_hook_the_checkout_wins():
    """Sanity of the test itself: `python script.py` puts the script directory first, so PYTHONPATH shadowing cannot work."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        _stub_checkout(tmp)
        r = subprocess.run([sys.executable, os.path.join(tmp, "inference.py")], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONPATH=ROOT), cwd="/", timeout=300)
        assert r.returncode != 0 and "must be shadowed" in r.stderr


REAL_CHECKOUT = "/root/reference"
REAL_DRIVER = """
import os, sys, warnings
sys.path.insert(0, {root!r})
from tools import refshim
refshim.install(third_party_only=True)        # timm / yacs / cv2 / torchvision stand-ins: the user's environment, not the checkout
import nmrf_amd.dropin as D
D.install()
sys.path.insert(0, {ref!r})                   # what `python inference.py` does with the script directory
warnings.simplefilter("error", ImportWarning)
for script, lo, hi in (("inference.py", 7, 12), ("main.py", 14, 19)):
    src = "\\n".join(open(os.path.join({ref!r}, script)).read().splitlines()[lo - 1:hi])
    assert "build_model" in src, src
    ns = {{}}
    exec(compile(src, script, "exec"), ns)
    print(script, ns["build_model"].__module__)
import nmrf.config, nmrf.utils.frame_utils as fu
print("config", nmrf.config.__file__)
print("frame_utils", fu.__file__, hasattr(fu, "InputPadder"), fu.downsample_disp.__module__)
cfg = ns["get_cfg"]() if "get_cfg" in ns else nmrf.config.get_cfg()
cfg.freeze()
model, criterion = ns["build_model"](cfg)
print("model", type(model).__module__, type(criterion).__module__, sorted(criterion.weight_dict)[-1], sum(p.numel() for p in model.parameters()))
import nmrf_amd.models.nmp as nmp, importlib
print("identity", nmp.__spec__.name, nmp.__package__)
importlib.reload(nmp)
"""


def test_real_checkout_import_lines_resolve_through_the_hook():
    """Build container only (the reference never travels: skipped where /root/reference is absent).  The actual import lines of
    the reference's inference.py:7-12 and main.py:14-19 are executed against the REAL checkout with the hook installed:
    nmrf.models / ops come from nmrf_amd, nmrf.config / nmrf.data / nmrf.utils from the checkout, build_model(cfg) takes the
    checkout's own yacs config and returns this build's NMRF (6 113 210 parameters); the aliased modules keep their
    own __spec__ (no ImportWarning, importlib.reload works)."""
    import pytest
    if not os.path.isfile(os.path.join(REAL_CHECKOUT, "inference.py")):
        pytest.skip("no reference checkout on this machine")
    r = subprocess.run([sys.executable, "-c", REAL_DRIVER.format(root=ROOT, ref=REAL_CHECKOUT)], capture_output=True, text=True,
                       cwd="/", timeout=300, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    assert out["inference.py"] == "nmrf_amd.models" and out["main.py"] == "nmrf_amd.models"
    assert out["config"] == os.path.join(REAL_CHECKOUT, "nmrf", "config", "__init__.py")
    assert out["frame_utils"] == "%s True nmrf_amd.frame_utils" % os.path.join(REAL_CHECKOUT, "nmrf", "utils", "frame_utils.py")
    assert out["model"] == "nmrf_amd.models.nmrf nmrf_amd.models.criterion proposal_disp 6113210"
    assert out["identity"] == "nmrf_amd.models.nmp nmrf_amd.models"
