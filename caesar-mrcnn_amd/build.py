"""In-tree build of the gfx950 kernel library (libmrcnn_hip.so) with hipcc.

No torch extension machinery: every .hip file under csrc/ is compiled to an object with
``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU) and linked into one shared library that
exports the C-ABI of include/mrcnn_hip.h.  Objects are rebuilt only when a source or header is newer.
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(PKG_DIR, "libmrcnn_hip.so")
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 kernel library")
    return exe


def _isa_check():
    """isa_check.py next to this file (build.py is also loaded by path, outside the package: __graft_entry__.build)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mrcnn_isa_check", os.path.join(PKG_DIR, "isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True, extra_flags=()):
    """Compile + link under an exclusive file lock (several processes may call this at once: every rank of a torchrun
    job on a fresh checkout); the library is linked to a temporary name and renamed into place, so a concurrent
    CDLL never sees a partial file."""
    import fcntl
    os.makedirs(os.path.join(PKG_DIR, "build"), exist_ok=True)
    with open(os.path.join(PKG_DIR, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, extra_flags)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, extra_flags):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h")))
    if not srcs:
        raise RuntimeError("no kernel sources under %s" % CSRC)
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = _newest(hdrs) if hdrs else 0.0
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-I", INCLUDE, "-I", CSRC,
             "-Wno-unused-result", "-ffp-contract=off"] + list(extra_flags)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        stale = force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time)
        if stale:
            jobs.append([_hipcc()] + flags + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if verbose or rc:
                    sys.stderr.write("[build] %s\n%s" % (os.path.basename(cmd[-3]), out))
                if rc:
                    raise RuntimeError("hipcc failed for %s:\n%s" % (cmd[-3], out[-4000:]))
        # machine-code check of what was just compiled (isa_check.py: wide buffer stores vs. vector writes of their data)
        isa_check = _isa_check()
        if isa_check.tools_available():
            for cmd in jobs:
                try:
                    isa_check.check_object(cmd[-1])
                except RuntimeError:
                    os.remove(cmd[-1])                     # never link (or skip as up to date) an object that failed the check
                    raise
    if jobs or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < _newest(objs):
        tmp = LIB_PATH + ".tmp.%d" % os.getpid()
        cmd, rc, out = run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs)
        if rc:
            sys.stderr.write(out)
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link of libmrcnn_hip.so failed:\n%s" % out[-4000:])
        # hipcc can silently drop the host stub of a kernel template (seen with a dependent array bound captured by a
        # lambda): the link succeeds, the library then fails to load.  Catch it here, with the name.
        nm = shutil.which("nm")
        if nm:
            und = subprocess.run([nm, "-D", "--undefined-only", tmp], capture_output=True, text=True).stdout
            missing = [l.split()[-1] for l in und.splitlines() if "__device_stub__" in l]
            if missing:
                os.remove(tmp)
                raise RuntimeError("kernel host stubs missing after link (compiler dropped them): %s" % ", ".join(missing[:4]))
        os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
