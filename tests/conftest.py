import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built libraries that do not come from the sources in the tree (a stale tools library once cost a GPU round)
    are rebuilt incrementally, and refused if that is not possible: both libraries carry a hash of every source they were built from
    (nmrf_build_stamp, python -m nmrf_amd.build)."""
    from nmrf_amd import build
    want = "abi%d-%s" % (__import__("nmrf_amd._lib", fromlist=["x"]).ABI_VERSION, build.source_stamp())
    libs = ((build.LIB, False), (build.LIB.replace("libnmrf_hip.so", "libnmrf_hip_debug.so"), True))
    stale = [(p, dbg) for p, dbg in libs if os.path.exists(p) and build.library_stamp(p) != want]
    for path, dbg in stale:                                  # an incremental rebuild first (seconds when little changed; hipcc needs no GPU)
        try:
            build.build_library(verbose=False, debug=dbg)
        except Exception as e:                               # no hipcc here: fall through to the refusal below
            sys.stderr.write("[conftest] rebuild of %s failed: %s\n" % (os.path.basename(path), str(e).splitlines()[0] if str(e) else repr(e)))
    for path, _ in stale:
        got = build.library_stamp(path)
        if got != want:
            pytest.exit("%s is stale: built from %s, the tree is %s -- run `python -m nmrf_amd.build`" % (
                os.path.relpath(path, ROOT), got, want), returncode=3)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- end-to-end disparity statistics: collected by the parity tests, printed after the run (and appended to
# gpurun_out/e2e_stats.jsonl on the GPU box) so that the numbers behind every green gate are in the pytest output
_DISP_STATS = []


def record_disp_stats(tag, stats):
    _DISP_STATS.append((tag, dict(stats)))
    try:
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "e2e_stats.jsonl"), "a") as f:
            f.write(json.dumps(dict(stats, case=tag)) + "\n")
    except OSError:
        pass


_NOTES = []


def record_note(text):
    _NOTES.append(text)


def pytest_terminal_summary(terminalreporter):
    for n in _NOTES:
        terminalreporter.write_line("note: " + n)
    if not _DISP_STATS:
        return
    terminalreporter.write_sep("-", "disparity agreement (px): EPE / EPE of non-flipped pixels / median / p99 / frac>0.5px / max")
    for tag, s in _DISP_STATS:
        terminalreporter.write_line("%-46s %.2e  %.2e  %.2e  %.2e  %.2e  %.3f" % (
            tag, s["epe"], s.get("epe_inliers", float("nan")), s["median"], s["p99"], s["frac_gt_0p5"], s["max"]))
