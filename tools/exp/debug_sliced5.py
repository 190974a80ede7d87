"""Which kernel is the victim?  Sliced evaluations with the tile-per-workgroup edge kernel forced (ARREAU_EDGE_SPLIT=1):
for every evaluation whose scores differ from the whole-batch reference, are the neighbour lists (returned edges) equal?"""
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
frac, types, lengths, angles, na = random_state(90, counts, 12, sampler_like=True)
B, N = len(counts), sum(counts)
d = lambda v: v.to(dev).contiguous()
off = crystal_offsets(na, dev)
t_c = torch.full((B,), 700, device=dev, dtype=torch.int32)
args = (d(frac), d(types.to(torch.int32)), d(lengths), d(angles), t_c, off)
eng.set_batch_layout(na, groups=1)
ref = eng.predict_scores(*args, return_edges=True)
ref_s = [x.clone() for x in ref[:3]]
ref_e = [x.clone() for x in ref[3]]
os.environ["ARREAU_SLICE_EAGER"] = "1"
eng.set_batch_layout(na, groups=2)
bad = edges_bad = 0
for it in range(int(os.environ.get("ITERS", "200"))):
    out = eng.predict_scores(*args, return_edges=True)
    torch.cuda.synchronize()
    s_eq = all(torch.equal(a, b) for a, b in zip(ref_s, out[:3]))
    e_eq = all(torch.equal(a, b) for a, b in zip(ref_e, out[3]))
    bad += not s_eq
    edges_bad += not e_eq
    if not s_eq and bad <= 3:
        which = [n for n, a, b in zip(("deg", "src", "dir", "dist"), ref_e, out[3]) if not torch.equal(a, b)]
        print("iteration", it, "scores differ; edge arrays that differ:", which)
        rdeg, rsrc, rdir, rdist = [x.cpu() for x in ref_e]
        odeg, osrc, odir, odist = [x.cpu() for x in out[3]]
        atoms = ((rdist.reshape(N, -1) != odist.reshape(N, -1)).any(1) | (rsrc.reshape(N, -1) != osrc.reshape(N, -1)).any(1)).nonzero().flatten().tolist()
        offs = off.cpu().tolist()
        for a in atoms[:4]:
            b = max(i for i in range(B) if offs[i] <= a)
            print("  atom", a, "crystal", b, "atoms in crystal", offs[b + 1] - offs[b], "deg", int(rdeg[a]), int(odeg[a]))
            print("    ref src ", rsrc.reshape(N, -1)[a].tolist(), "dist", [round(float(v), 6) for v in rdist.reshape(N, -1)[a]])
            print("    out src ", osrc.reshape(N, -1)[a].tolist(), "dist", [round(float(v), 6) for v in odist.reshape(N, -1)[a]])
        print("  atoms affected:", atoms)
print("score mismatches", bad, "edge-array mismatches", edges_bad)
