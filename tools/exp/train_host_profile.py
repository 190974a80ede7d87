"""Host-side profile of the training step (cProfile over 20 steps of bench.py's config-5 loop on one GPU)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd import build
build.build(verbose=False)
from arreau_amd.checkpoint import default_args
from arreau_amd.diffusion.lattice_dataset import CrystalDataset, collate, synthetic_alexandria_like
from arreau_amd.lightning_wrappers.diffusion import PONITA_DIFFUSION
from arreau_amd.train import optimizer_step

dev = torch.device("cuda", 0)
ds = CrystalDataset(configs=synthetic_alexandria_like(4096, seed=0))
torch.manual_seed(1234)
model = PONITA_DIFFUSION(default_args(lr=3e-4, epochs=10, hidden_dim=int(os.environ.get("HIDDEN_DIM", "128"))), ds.z_table).to(dev)
optimizer = model.configure_optimizers(max_epochs=10)["optimizer"]
rng = np.random.RandomState(100)
batches = [collate([ds[int(i)] for i in rng.choice(len(ds), 64, replace=False)]) for _ in range(8)]


def step(i):
    model.training_step(batches[i % 8])
    optimizer_step(model, optimizer, 1)


for i in range(4):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue time {1e3 * t_host / 20:.2f} ms/step, with final sync {1e3 * t_all / 20:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
