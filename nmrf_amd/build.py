"""Build libnmrf_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to compile.

    python -m nmrf_amd.build            # incremental
    python -m nmrf_amd.build --force
    python -m nmrf_amd.build --debug    # tools-only libnmrf_hip_debug.so (probes, micro-benchmarks)
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnmrf_hip.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libnmrf_hip.so)")


# kernels the product no longer launches: reference paths for A/B parity runs and micro-benchmarks, compiled into the tools / test
# build only (include/nmrf_hip_debug.h)
DEBUG_ONLY = ("token_linear.hip", "nmp_block.hip", "conv_wino.hip")


def _sources(debug=False):
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and (debug or f not in DEBUG_ONLY))


def source_stamp():
    """sha256/16 over every source and header a library is built from (names + contents, both libraries hash the same set: the
    tools library is a superset build of the same tree)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and not f.startswith("_ab_"))
    files += [os.path.join(os.path.dirname(HERE), "include", f) for f in ("nmrf_hip.h", "nmrf_hip_debug.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def library_stamp(path):
    """nmrf_build_stamp() of a built library ("abi<N>-<hash>") read from the file's bytes, or None.  Deliberately NOT through dlopen: a
    HIP library loaded before `import torch` binds to the system's libamdhip64 while torch brings its own -- every later launch from
    that library then fails (found the hard way: a conftest hook that dlopen-ed the libraries first, round 6)."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(rb"abi\d+-[0-9a-f]{16}(?=\x00)", f.read())
        return m.group(0).decode() if m else None
    except OSError:
        return None


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "nmrf_hip.h"))
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "nmrf_hip_debug.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build_library(force=False, verbose=True, debug=False):
    """debug=True builds libnmrf_hip_debug.so: the same sources with -DNMRF_DEBUG_PROBES (nmrf_debug_* entry points, peak
    micro-benchmarks, census / stamp instantiations, tuning getenvs).  Used by tools/ only; the product never loads it."""
    hipcc = _hipcc()
    lib_path = os.path.join(LIBDIR, "libnmrf_hip_debug.so") if debug else LIB
    objdir = OBJDIR + ("_debug" if debug else "")
    flags = FLAGS + (["-DNMRF_DEBUG_PROBES"] if debug else [])
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _deps_mtime()
    jobs = []
    objs = []
    stamp = source_stamp()
    stamp_file = os.path.join(objdir, "build_stamp.txt")
    old_stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
    for src in _sources(debug):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
        extra = []
        if os.path.basename(src) == "build_stamp.hip":            # carries the hash of the whole tree: rebuilt whenever that changes
            stale = stale or old_stamp != stamp
            extra = ['-DNMRF_BUILD_STAMP="%s"' % stamp]
        if stale:
            jobs.append([hipcc] + flags + extra + ["-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, rc, log in ex.map(run, jobs):
                if verbose:
                    print("[nmrf_amd.build]", os.path.basename(cmd[-3]), "rc=%d" % rc)
                if rc != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), log))
    with open(stamp_file, "w") as f:
        f.write(stamp + "\n")
    stale_lib = not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs)
    if jobs or stale_lib:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs
        cmd, rc, log = run(cmd)
        if rc != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), log))
        if verbose:
            print("[nmrf_amd.build] linked", lib_path)
    return lib_path


if __name__ == "__main__":
    # both libraries by default: a product build next to a stale tools build fails every A/B test with an ABI mismatch
    # (--debug / --main-only: one of them)
    if "--debug" not in sys.argv:
        build_library(force="--force" in sys.argv)
    if "--main-only" not in sys.argv:
        build_library(force="--force" in sys.argv, debug=True)
