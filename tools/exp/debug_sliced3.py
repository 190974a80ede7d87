import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd.checkpoint import make_synthetic_model
from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
from tests.helpers import random_state
dev = torch.device("cuda", 0)
m = make_synthetic_model(S=90, seed=1234).to(dev)
eng = m.engine()
rng = np.random.RandomState(3)
counts = [int(v) for v in rng.randint(3, 21, size=37)]
if os.environ.get("CRYSTALS"):  # uniform batch of a BASELINE configuration instead of the ragged test batch
    counts = [int(os.environ.get("ATOMS", "20"))] * int(os.environ["CRYSTALS"])
frac, types, lengths, angles, na = random_state(90, counts, 12, sampler_like=True)
B, N = len(counts), sum(counts)
d = lambda v: v.to(dev).contiguous()
off = crystal_offsets(na, dev)
G = int(os.environ.get("GROUPS", "2"))
steps = int(os.environ.get("STEPS", "5"))
def loop(use_graph):
    f, ty, le, lat = d(frac.clone()), d(types.to(torch.int32)), d(lengths.clone()), torch.zeros(B, 3, 3, device=dev)
    eng.sample_loop(f, ty, le, d(angles), off, 999, steps, 4242, None, lat, use_graph=use_graph)
    torch.cuda.synchronize()
    return f, ty, le, lat
eng.set_batch_layout(na, groups=1)
ref = loop(False)
ref2 = loop(False)
print("eager whole repeat:", [bool(torch.equal(a, b)) for a, b in zip(ref, ref2)])
eng.set_batch_layout(na, groups=G)
offs = off.cpu().numpy()
for use_graph in (os.environ.get("EAGER") is None,) * int(os.environ.get("RUNS", "6")):
    out = loop(use_graph)
    msg = []
    for name, a, b in zip(("frac", "types", "lengths", "lattice"), ref, out):
        if not torch.equal(a, b):
            diff = (a.float() - b.float()).abs().reshape(a.shape[0], -1).max(1).values
            rows = (diff > 0).nonzero().flatten().tolist()
            msg.append((name, len(rows), rows[:6], float(diff.max())))
    print("use_graph", use_graph, "diffs:", [(mm[0], mm[1], mm[3]) for mm in msg])
