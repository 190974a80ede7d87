"""Host-side profile of the training step (bench.py --config c5's loop): where does the HOST spend its time per step?"""
import cProfile
import os
import pstats
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
for i in range(6):
    t0 = time.perf_counter(); model.training_step(batches[i % 8]); t1 = time.perf_counter(); optimizer_step(model, optimizer, 1); t2 = time.perf_counter()
    print("step %d: training_step %.2f ms, optimizer_step %.2f ms (host)" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    model.training_step(batches[i % 8]); optimizer_step(model, optimizer, 1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45); pstats.Stats(pr).sort_stats("tottime").print_stats(25)
