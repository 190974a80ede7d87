"""Training data path (BASELINE config 5): the HDF5 dataset layout of the reference and the in-memory
`CrystalDataset`, without h5py / torch_geometric / pymatgen (none is installed in this image).

Layout written by diffusion/prep_datasets.py:67-79 and read by diffusion/lattice_dataset.py:23-42:
    atomic_number/<i>   int   [n_i]     atomic numbers of crystal i (NOT one-hot)
    lattice_matrix      float [M, 3, 3]
    frac_coord/<i>      float [n_i, 3]
HDF5 when h5py is importable; otherwise an `.npz` whose keys are exactly those paths.

`CrystalDataset.__getitem__` returns the fields of the reference's PyG `Data` (lattice_dataset.py:96-104: pos, X0, A0,
L0, num_atoms) and `collate` concatenates them the way PyG's DataLoader does (node tensors along dim 0, so L0 becomes
[B*3, 3] -- DiffusionLoss.__call__ views it back to [B, 3, 3], diffusion_loss.py:207-208 -- plus the `batch` index).
The Alexandria files are not available offline; `synthetic_alexandria_like` draws crystals with the dataset's published
statistics (mean about 8 atoms per cell, density 0.0554 atoms / A^3) for tests and the config-5 bench.
"""
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch

from .tools.atomic_number_table import atomic_numbers_to_indices, get_atomic_number_table_from_zs


@dataclass
class Configuration:
    atomic_numbers: np.ndarray  # [n] atomic numbers, not one-hot
    X0: np.ndarray              # [n, 3] fractional coordinates
    L0: np.ndarray              # [3, 3] cell rows


def _is_h5(filename):
    return str(filename).endswith((".h5", ".hdf5"))


def save_dataset(filename: str, atomic_number_vectors: Sequence[np.ndarray], lattice_matrices: np.ndarray,
                 frac_coords_arrays: Sequence[np.ndarray]):
    """prep_datasets.py:67-79."""
    lattice_matrices = np.asarray(lattice_matrices, dtype=np.float64)
    if not (len(atomic_number_vectors) == len(frac_coords_arrays) == lattice_matrices.shape[0]):
        raise ValueError("one atomic-number vector, one cell and one coordinate array per crystal")
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    if _is_h5(filename):
        import h5py
        with h5py.File(filename, "w") as f:
            g = f.create_group("atomic_number")
            for i, v in enumerate(atomic_number_vectors):
                g.create_dataset(str(i), data=np.asarray(v, dtype=int))
            f.create_dataset("lattice_matrix", data=lattice_matrices)
            g = f.create_group("frac_coord")
            for i, a in enumerate(frac_coords_arrays):
                g.create_dataset(str(i), data=np.asarray(a, dtype=np.float64))
    else:
        arrays = {"lattice_matrix": lattice_matrices}
        for i, v in enumerate(atomic_number_vectors):
            arrays[f"atomic_number/{i}"] = np.asarray(v, dtype=int)
        for i, a in enumerate(frac_coords_arrays):
            arrays[f"frac_coord/{i}"] = np.asarray(a, dtype=np.float64)
        np.savez(filename, **arrays)
    return filename


def load_data(filename: str):
    """lattice_dataset.py:23-42: (atomic_number_vectors, lattice_matrices, frac_coords_arrays), groups in numeric order."""
    if _is_h5(filename):
        import h5py
        with h5py.File(filename, "r") as f:
            zs = [np.array(f["atomic_number"][k]) for k in sorted(f["atomic_number"], key=int)]
            lattice = np.array(f["lattice_matrix"])
            fc = [np.array(f["frac_coord"][k]) for k in sorted(f["frac_coord"], key=int)]
    else:
        with np.load(filename) as z:
            zk = sorted((k for k in z.files if k.startswith("atomic_number/")), key=lambda k: int(k.split("/")[1]))
            fk = sorted((k for k in z.files if k.startswith("frac_coord/")), key=lambda k: int(k.split("/")[1]))
            zs = [z[k] for k in zk]
            lattice = z["lattice_matrix"]
            fc = [z[k] for k in fk]
    return zs, lattice, fc


def load_dataset(file_path) -> List[Configuration]:
    """lattice_dataset.py:45-57."""
    zs, lattice, fc = load_data(file_path)
    out = []
    for i in range(len(lattice)):
        assert lattice[i].shape == (3, 3)
        out.append(Configuration(atomic_numbers=zs[i], X0=fc[i], L0=lattice[i]))
    return out


class CrystalDataset(torch.utils.data.Dataset):
    """lattice_dataset.py:73-113 (all files loaded into memory; the class table is built from the atomic numbers that
    occur, plus the mask state)."""

    def __init__(self, config_paths: Optional[Sequence[str]] = None, cutoff: float = 5.0,
                 configs: Optional[List[Configuration]] = None):
        if configs is None:
            configs = [c for p in config_paths for c in load_dataset(p)]
        self.configs = list(configs)
        self.cutoff = cutoff
        self.unique_atomic_numbers = set()
        for c in self.configs:
            self.unique_atomic_numbers.update(int(z) for z in c.atomic_numbers)
        self.z_table = get_atomic_number_table_from_zs([self.unique_atomic_numbers])

    def __len__(self):
        return len(self.configs)

    def __getitem__(self, idx: int):
        dt = torch.float64  # the reference fixes float64 here (lattice_dataset.py:88)
        c = self.configs[idx]
        return SimpleNamespace(pos=torch.tensor(np.asarray(c.X0) @ np.asarray(c.L0), dtype=dt),
                               X0=torch.tensor(np.asarray(c.X0), dtype=dt),
                               A0=torch.as_tensor(atomic_numbers_to_indices(self.z_table, c.atomic_numbers), dtype=torch.long),
                               L0=torch.tensor(np.asarray(c.L0), dtype=dt), num_atoms=len(c.atomic_numbers))


def collate(items):
    """What torch_geometric's DataLoader makes of a list of those `Data` objects: tensors concatenated along dim 0,
    python ints stacked, plus `batch` (crystal of each atom) and `ptr`."""
    n = torch.tensor([int(it.num_atoms) for it in items], dtype=torch.long)
    B = len(items)
    return SimpleNamespace(pos=torch.cat([it.pos for it in items]), X0=torch.cat([it.X0 for it in items]),
                           A0=torch.cat([it.A0 for it in items]), L0=torch.cat([it.L0 for it in items]), num_atoms=n,
                           batch=torch.arange(B).repeat_interleave(n), ptr=torch.cat([n.new_zeros(1), n.cumsum(0)]),
                           num_graphs=B)


def iterate_batches(dataset, batch_size: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1,
                    drop_last: bool = False):
    """One epoch of batches for data-parallel rank `rank` of `world_size`: every rank uses the same permutation (seeded)
    and takes every world_size-th crystal, like DistributedSampler under Lightning's DDP (main_diffusion.py:293-303).
    With several ranks the permutation is first cut to a multiple of world_size (DistributedSampler's drop_last), so every
    rank sees the same number of crystals and therefore of batches: a rank with one batch more than its peers would wait
    for ever in that step's all-reduce."""
    order = np.arange(len(dataset))
    if shuffle:
        np.random.RandomState(seed).shuffle(order)
    if world_size > 1:
        order = order[:len(order) - len(order) % world_size]
    mine = order[rank::world_size]
    for s in range(0, len(mine), batch_size):
        idx = mine[s:s + batch_size]
        if drop_last and len(idx) < batch_size:
            break
        yield collate([dataset[int(i)] for i in idx])


def synthetic_alexandria_like(num_crystals: int, seed: int = 0, num_species: int = 89, max_atoms: int = 64,
                              density: float = 0.05539856) -> List[Configuration]:
    """Crystals with Alexandria-PBE's published statistics (exploration/find_avg_density_of_dataset.py:40-41: 0.0554
    atoms / A^3, mean cell volume 152.5 A^3, i.e. about 8 atoms per cell; capped at `max_atoms`): atom counts from a
    geometric-like law with mean 8, X0 ~ U[0,1), a random well-conditioned triclinic cell scaled to the density,
    atomic numbers uniform in 1..num_species."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(num_crystals):
        n = int(min(max_atoms, 1 + rng.geometric(1.0 / 7.5)))
        lengths = rng.uniform(0.7, 1.4, size=3)
        ang = np.deg2rad(rng.uniform(75, 105, size=3))
        ca, cb, cg = np.cos(ang)
        sa, sb = np.sin(ang[0]), np.sin(ang[1])
        gs = np.arccos(np.clip((ca * cb - cg) / (sa * sb), -1, 1))
        L = np.array([[lengths[0] * sb, 0.0, lengths[0] * cb],
                      [-lengths[1] * sa * np.cos(gs), lengths[1] * sa * np.sin(gs), lengths[1] * ca],
                      [0.0, 0.0, lengths[2]]])
        L *= (n / density / abs(np.linalg.det(L))) ** (1.0 / 3.0)
        out.append(Configuration(atomic_numbers=rng.randint(1, num_species + 1, size=n), X0=rng.uniform(0, 1, size=(n, 3)),
                                 L0=L))
    return out
