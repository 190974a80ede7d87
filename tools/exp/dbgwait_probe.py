"""Which outputs differ between the product library and its debug-wait twin (every counted wait -> vmcnt(0)), and by how much?
Runs the evaluation of tests/test_gpu_parity.py::test_launch_geometry_switches... in subprocesses: library x ARREAU_CROSS_FP8, twice each."""
import os, subprocess, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = (
    "import torch, sys; sys.path.insert(0, %r)\n"
    "from arreau_amd.checkpoint import make_synthetic_model\n"
    "from arreau_amd.diffusion.diffusion_helpers import crystal_offsets\n"
    "from tests.helpers import random_state\n"
    "dev = torch.device('cuda', 0)\n"
    "m = make_synthetic_model(S=90, seed=1234).to(dev)\n"
    "frac, types, lengths, angles, na = random_state(90, [20] * 96, 5, sampler_like=True)\n"
    "d = lambda v: v.to(dev).contiguous()\n"
    "t_c = torch.full((96,), 500, device=dev, dtype=torch.int32)\n"
    "out = m.engine().predict_scores(d(frac), d(types.to(torch.int32)), d(lengths), d(angles), t_c, crystal_offsets(na, dev), return_edges=True)\n"
    "torch.save([x.cpu() for x in out[:3]] + [out[3][0].cpu()], sys.argv[1])\n" % ROOT)
sys.path.insert(0, ROOT)
from arreau_amd.build import LIB, LIB_DEBUG_WAIT
outs = {}
with tempfile.TemporaryDirectory() as d:
    for lib in ("prod", "dbg"):
        for x8 in ("1", "0"):
            for rep in (0, 1):
                env = {**os.environ, "ARREAU_BASIS_MIN_RECEIVERS": "240", "ARREAU_CROSS_FP8": x8}
                if lib == "dbg":
                    env["ARREAU_HIP_LIB"] = LIB_DEBUG_WAIT
                p = os.path.join(d, "o.pt")
                subprocess.run([sys.executable, "-c", code, p], check=True, env=env, timeout=300)
                outs[(lib, x8, rep)] = torch.load(p)
def cmp(a, b):
    return " ".join("%s: %d differ, max %.2e" % (n, int((x != y).sum()), float((x - y).abs().max())) for n, x, y in zip(("eps", "logits", "len0"), a, b))
for x8 in ("1", "0"):
    print("x8=%s prod run0 vs run1:" % x8, cmp(outs[("prod", x8, 0)], outs[("prod", x8, 1)]))
    print("x8=%s dbg  run0 vs run1:" % x8, cmp(outs[("dbg", x8, 0)], outs[("dbg", x8, 1)]))
    print("x8=%s prod vs dbg      :" % x8, cmp(outs[("prod", x8, 0)], outs[("dbg", x8, 0)]))
    a, b = outs[("prod", x8, 0)], outs[("dbg", x8, 0)]
    bad = ((a[0] != b[0]).any(1) | (a[1] != b[1]).any(1)).nonzero().flatten()
    print("   atoms that differ:", bad[:20].tolist(), "of", a[0].shape[0], "; their in-degrees:", a[3][bad[:20]].tolist())
print("prod x8=1 vs x8=0:", cmp(outs[("prod", "1", 0)], outs[("prod", "0", 0)]))
