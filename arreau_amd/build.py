"""Build recipe for libarreau_hip.so (gfx950 only; hipcc cross-compiles without a GPU).

    python -m arreau_amd.build [--force]

The library is built IN-TREE (arreau_amd/csrc/libarreau_hip.so) so it travels with the source
snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys
import time

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["model.hip", "graph.hip", "edge.hip", "edge_bf16.hip", "edge_f16.hip", "node.hip", "node_bf16.hip", "node_f16.hip", "node_f16m.hip", "update.hip", "api.hip"]
HEADERS = ["internal.h", "bf16x6.h", "f16x3.h", os.path.join("..", "..", "include", "arreau_hip.h")]
LIB = os.path.join(CSRC, "libarreau_hip.so")
STAMP = os.path.join(CSRC, ".build_stamp")
# Sources whose kernels hand-count s_waitcnt vmcnt(N): a register spill would put scratch loads/stores into the same
# in-order queue and silently break the count, so the build fails if the compiler reports any scratch for them.
NO_SCRATCH = {"edge_f16.hip"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"] + os.environ.get("ARREAU_EXTRA_HIPCC_FLAGS", "").split()


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _digest()


def build(force=False, verbose=True):
    """Compile every .hip source to an object (in parallel) and link the shared library.
    Serialised across processes with a file lock (one rank builds, the others find the result)."""
    if not force and not needs_build():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build_lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    t0 = time.time()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src in NO_SCRATCH:
            cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode == 0 and src in NO_SCRATCH:
            import re
            scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out)]
            if not scratch or max(scratch) != 0:
                failed = True
                sys.stderr.write(f"[arreau_amd.build] {src}: kernel uses scratch {scratch} (register spill); "
                                 "its counted vmcnt waits require none\n")
            out = ""  # the remarks are not warnings
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[arreau_amd.build] {src} failed:\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(f"[arreau_amd.build] {src}:\n{out}\n")
    if failed:
        raise RuntimeError("hipcc failed (see messages above)")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    if verbose:
        sys.stderr.write(f"[arreau_amd.build] built {LIB} in {time.time() - t0:.1f}s\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
