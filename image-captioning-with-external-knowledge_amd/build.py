"""hipcc recipe: csrc/*.hip -> libick_amd.so for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libick_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-Wall",
         "-Wno-unused-function",
         "-ffp-contract=off"]  # fused multiply-adds only where the source says fmaf / MFMA


def source_id():
    """16 hex digits over everything that shapes the timed step: the kernel sources, the C header and the host modules
    that schedule the launches.  Measurement tables made from profiler runs (profiles/*_in_step_*.json, *_traffic_*.json)
    carry the id they were made with; bench.py marks them stale when it differs from the tree it runs from."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(INCLUDE, "ick_amd.h")] + [os.path.join(PKG, f) for f in ("training.py", "weights.py", "decoder.py", "ops.py", "dp.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    # a change of compiler flags invalidates every object
    stamp = os.path.join(CSRC, ".flags")
    cur = " ".join([HIPCC] + [f for f in FLAGS if not os.path.isabs(f)])   # location independent: objects may be
                                                                            # built in another checkout of the tree
    if not os.path.exists(stamp) or open(stamp).read() != cur:
        force = True
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "ick_amd.h")]
    objs, jobs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn)
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs)
    open(stamp, "w").write(cur)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
