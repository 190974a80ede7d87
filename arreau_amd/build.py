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
SOURCES = ["model.hip", "graph.hip", "edge.hip", "edge_bf16.hip", "edge_f16.hip", "node.hip", "node_bf16.hip", "node_f16m.hip", "conv_proj.hip", "update.hip", "train.hip", "train_net.hip", "optim.hip", "api.hip"]
HEADERS = ["internal.h", "bf16x6.h", "f16x3.h", os.path.join("..", "..", "include", "arreau_hip.h"), "sgemm.h", "philox.h", "embed_dev.h", "prep_dev.h", "graph_dev.h", "update_dev.h", "readout_dev.h"]
LIB = os.path.join(CSRC, "libarreau_hip.so")
# Debug twin: the same sources with -DARREAU_DEBUG_WAIT_ALL (every hand-counted `s_waitcnt vmcnt(N)` becomes vmcnt(0)).
# Its outputs must be bit-identical to the product library's (test_counted_waits_match_full_waits); only the sources
# that contain counted waits are recompiled for it.
LIB_DEBUG_WAIT = os.path.join(CSRC, "libarreau_hip_dbgwait.so")
DEBUG_WAIT_SOURCES = ["edge_f16.hip", "node.hip", "node_f16m.hip", "conv_proj.hip"]
STAMP = os.path.join(CSRC, ".build_stamp")
# Kernels that hand-count s_waitcnt vmcnt(N) or drain LDS-DMA copies with asm waits hipcc cannot see: a register spill
# would put scratch loads/stores into the same in-order queue and silently break the protocol, so the build fails if
# the compiler reports scratch for them.  source -> substrings of the (mangled) kernel names to check (None = all).
# (conv_proj.hip is covered by the ISA lint instead: its projection role may spill -- it issues no asm memory operation
# and hipcc counts its own scratch traffic --, its mix role, which hand-counts, must not share a path with a spill)
NO_SCRATCH = {"edge_f16.hip": None, "node.hip": ["conv_kernel_streamed"], "node_f16m.hip": None}
# Sources with inline asm: their device ISA is kept (-save-temps) and run through arreau_amd/_isa_lint.py -- software wait
# states around every asm instruction (store-data, VALU-written SGPR -> VMEM, M0 -> LDS-DMA, ...), asm loads' destination
# registers untouched until their wait, no compiler use of M0, no unmodelled instruction kind inside asm.  hipcc pads and
# counts none of that for inline asm; a violation fails the build.
ASM_LINT = ("edge_f16.hip", "node.hip", "node_f16m.hip", "graph.hip", "api.hip", "conv_proj.hip", "train_net.hip")
# -Wno-inline-asm: the lean LDS-DMA asm lists "m0" as clobbered (it overwrites M0 and does not restore it); clang warns
# that reserved registers in a clobber list are not preserved for us -- which is what is declared, not asked for.  The
# ISA check below verifies that the compiler itself never uses M0 in those kernels.
# Extra flags per source.  The split-precision kernels keep their epilogue arithmetic as independent fp32 instructions
# (f16x3.h): the SLP vectoriser would fuse adjacent ones back into v_pk_*_f32.
PER_SOURCE_FLAGS = {src: ["-fno-slp-vectorize"] for src in ("edge_f16.hip", "node_f16m.hip")}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-Wno-unused-but-set-variable", "-Wno-misleading-indentation", "-ffp-contract=on"] + os.environ.get("ARREAU_EXTRA_HIPCC_FLAGS", "").split()


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [os.path.join("..", "_isa_lint.py")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_SOURCE_FLAGS.items())).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB_DEBUG_WAIT) or not os.path.exists(STAMP):
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


def _scratch_by_kernel(remarks):
    """{kernel name: scratch bytes per lane} from -Rpass-analysis=kernel-resource-usage output."""
    import re
    out, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name is not None:
            out[name] = int(m.group(1))
    return out


def _drop_remarks(diag):
    """Compiler output without the resource-usage remark blocks (remark line + quoted source + caret)."""
    import re
    keep, skipping = [], False
    for line in diag.splitlines():
        if re.match(r"^\S+:\d+:\d+: (warning|error|note|remark):", line) or re.match(r"^remark:", line):
            skipping = "remark:" in line
        elif re.match(r"^\d+ warnings? generated", line):
            continue
        if not skipping:
            keep.append(line)
    return "\n".join(keep)


def _isa_lint():
    """The ISA hazard lint (arreau_amd/_isa_lint.py; tools/isa_lint.py is its command-line wrapper)."""
    from arreau_amd import _isa_lint as mod
    return mod


def _compile_all(hipcc, sources, objdir, extra_flags, verbose):
    import glob
    import shutil
    import tempfile
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + PER_SOURCE_FLAGS.get(src, []) + extra_flags
        tmp = None
        if src in NO_SCRATCH or src in ASM_LINT:  # + keep the device ISA for the lint (temporaries go to a scratch directory)
            tmp = tempfile.mkdtemp(prefix="arreau_isa_")
            cmd += ["-Rpass-analysis=kernel-resource-usage", "-save-temps"]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, tmp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=tmp)))
    failed = False
    for src, tmp, p in procs:
        out, _ = p.communicate()
        if tmp is not None:
            if p.returncode == 0 and src in ASM_LINT:
                isa = glob.glob(os.path.join(tmp, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
                lint = _isa_lint()
                bad = lint.lint_file(isa[0]) if isa else None
                if bad is None or bad:
                    failed = True
                    sys.stderr.write(f"[arreau_amd.build] {src}: ISA hazard lint (arreau_amd/_isa_lint.py): "
                                     f"{'device ISA not found' if bad is None else str(len(bad)) + ' violation(s)'}\n")
                    for v in (bad or [])[:8]:
                        sys.stderr.write("  " + lint.format_violation(v) + "\n")
                elif verbose:
                    nf, ni, na = lint.summarize(isa[0])
                    sys.stderr.write(f"[arreau_amd.build] {src}: ISA lint clean ({nf} kernels, {ni} instructions, {na} from inline asm)\n")
            shutil.rmtree(tmp, ignore_errors=True)
        if p.returncode == 0 and src in ASM_LINT and src not in NO_SCRATCH:
            out = _drop_remarks(out)
        if p.returncode == 0 and src in NO_SCRATCH:
            scratch = _scratch_by_kernel(out)
            want = NO_SCRATCH[src]
            checked = {k: v for k, v in scratch.items() if want is None or any(w in k for w in want)}
            bad = {k: v for k, v in checked.items() if v != 0}
            if not checked or bad:
                failed = True
                sys.stderr.write(f"[arreau_amd.build] {src}: kernels with hand-written wait protocols must not use scratch "
                                 f"(register spill); compiler reports {bad or 'no kernels'}\n")
            out = _drop_remarks(out)
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[arreau_amd.build] {src} failed:\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(f"[arreau_amd.build] {src}:\n{out}\n")
    if failed:
        raise RuntimeError("hipcc failed (see messages above)")
    return objs


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    t0 = time.time()
    objs = _compile_all(hipcc, SOURCES, CSRC, [], verbose)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    dbg_dir = os.path.join(CSRC, "dbgwait")
    dbg = _compile_all(hipcc, DEBUG_WAIT_SOURCES, dbg_dir, ["-DARREAU_DEBUG_WAIT_ALL"], verbose)
    dbg_objs = [os.path.join(dbg_dir, os.path.basename(o)) if os.path.basename(o).replace(".o", ".hip") in DEBUG_WAIT_SOURCES
                else o for o in objs]
    assert all(os.path.exists(o) for o in dbg_objs) and len(dbg) == len(DEBUG_WAIT_SOURCES)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_DEBUG_WAIT] + dbg_objs)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    if verbose:
        sys.stderr.write(f"[arreau_amd.build] built {LIB} (+ debug-wait twin) in {time.time() - t0:.1f}s\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
