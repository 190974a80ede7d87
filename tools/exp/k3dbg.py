import os, sys, torch
sys.path.insert(0, os.getcwd())
from arreau_amd.checkpoint import make_synthetic_model
from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
from tests.helpers import random_state
dev = torch.device('cuda', 0)
m = make_synthetic_model(S=90, seed=1234).to(dev)
for B, n in ((3, 7), (256, 20), (40, 20)):
    frac, types, lengths, angles, na = random_state(90, [n] * B, 100 + B, cell=(4.0, 8.0))
    d = lambda v: v.to(dev).contiguous()
    t_c = torch.full((B,), 999, device=dev, dtype=torch.int32)
    args = (d(frac), d(types.to(torch.int32)), d(lengths), d(angles), t_c, crystal_offsets(na, dev))
    outs = [m.engine().predict_scores(*args) for _ in range(3)]
    torch.cuda.synchronize()
    print(B, n, "K3=%s" % os.environ.get("ARREAU_K3", "1"), "repeat equal:", [bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[1])],
          "nan:", [int(torch.isnan(a).sum()) for a in outs[0]], "absmax:", [float(a.abs().max()) for a in outs[0]])
    torch.save([a.cpu() for a in outs[0]], "/tmp/k3_%s_%d_%d.pt" % (os.environ.get("ARREAU_K3", "1"), B, n))
