"""Sporadic host stalls of the training loop: per-step host times under a few variants (argv[1]: base | gc_off | sync | nobound)."""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd import build  # noqa: E402

build.build(verbose=False)
from arreau_amd.checkpoint import default_args  # noqa: E402
from arreau_amd.diffusion.lattice_dataset import CrystalDataset, collate, synthetic_alexandria_like  # noqa: E402
from arreau_amd.lightning_wrappers.diffusion import PONITA_DIFFUSION  # noqa: E402
from arreau_amd.train import optimizer_step  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda", 0)
ds = CrystalDataset(configs=synthetic_alexandria_like(4096, seed=0))
torch.manual_seed(1234)
model = PONITA_DIFFUSION(default_args(lr=3e-4, epochs=10, hidden_dim=128), ds.z_table).to(dev)
optimizer = model.configure_optimizers(max_epochs=10)["optimizer"]
rng = np.random.RandomState(100)
batches = [collate([ds[int(i)] for i in rng.choice(len(ds), 64, replace=False)]) for _ in range(8)]
model.diffusion_loss(model, max(batches, key=lambda b: int(b.num_atoms.sum())), None, training=True)
for i in range(5):
    model.training_step(batches[i % 8]); optimizer_step(model, optimizer, 1)
torch.cuda.synchronize()
if variant == "gc_off":
    gc.disable()
if variant == "gc_freeze":
    gc.collect(); gc.freeze()
ts, os_ = [], []
t_all = time.perf_counter()
for i in range(60):
    if variant == "nobound":
        model.__dict__.setdefault("_steps_in_flight", __import__("collections").deque()).clear()
    t0 = time.perf_counter(); model.training_step(batches[i % 8]); t1 = time.perf_counter(); optimizer_step(model, optimizer, 1); t2 = time.perf_counter()
    if variant == "sync":
        torch.cuda.synchronize()
    ts.append(1e3 * (t1 - t0)); os_.append(1e3 * (t2 - t1))
torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t_all) / 60
print("%-9s %.2f ms/step; host training_step median %.2f max %.1f; optimizer median %.2f max %.1f; stalls > 10 ms at steps %s" % (
    variant, tot, np.median(ts), max(ts), np.median(os_), max(os_), [i for i in range(60) if ts[i] > 10 or os_[i] > 10]), flush=True)
