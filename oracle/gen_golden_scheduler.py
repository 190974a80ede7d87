"""Test infrastructure: learning-rates the REFERENCE's CosineWarmupScheduler hands to an optimizer of lr = 1, epoch by epoch
(/root/reference/lightning_wrappers/scheduler.py, imported; SURVEY 8c lists the module as importable).  Writes
tests/golden/scheduler.json.  Run in the build container only (the reference does not travel to the GPU box)."""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
from lightning_wrappers.scheduler import CosineWarmupScheduler  # noqa: E402

cases = []
for warmup, max_iters, epochs in ((10, 100, 100), (0, 50, 50), (5, 5, 5), (3, 1000, 40)):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1.0)
    sch = CosineWarmupScheduler(opt, warmup, max_iters)
    lrs = []
    for _ in range(epochs + 1):
        lrs.append(float(opt.param_groups[0]["lr"]))
        opt.step()
        sch.step()
    cases.append({"warmup": warmup, "max_iters": max_iters, "lr_by_epoch": lrs})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scheduler.json")
with open(out, "w") as fh:
    json.dump(cases, fh)
print("wrote", out, sum(len(c["lr_by_epoch"]) for c in cases), "values")
