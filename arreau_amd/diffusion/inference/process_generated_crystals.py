"""Result wire format of the sampler: the `crystals` group of out/crystals.h5
(diffusion/inference/process_generated_crystals.py:8-47 of the reference) -- writer, reader and the
per-crystal slicing helpers, so `main_diffusion_process_results.py`, the notebook viewer and the MACE
relaxation of the reference keep working on files written here.

Layout (main_diffusion_generate.py:67-92): frac_x [sum n, 3] float64, atomic_numbers [sum n] float64 (the
reference fills an `np.empty` float array), lattice [B, 3, 3] float64, idx_start [B] int64 (first atom of
each crystal), num_atoms [B] int64.  HDF5 when h5py is importable (it is not in this image); `.npz`
with the same five keys otherwise."""
import os

import numpy as np

from ..diffusion_loss import SampleResult

KEYS = ("frac_x", "atomic_numbers", "lattice", "idx_start", "num_atoms")
_DTYPES = dict(frac_x=np.float64, atomic_numbers=np.float64, lattice=np.float64, idx_start=np.int64,
               num_atoms=np.int64)


def _fields(crystals: SampleResult):
    out = {}
    for k in KEYS:
        v = getattr(crystals, k)
        if v is None:
            raise ValueError(f"SampleResult.{k} is missing")
        out[k] = np.asarray(v).astype(_DTYPES[k], copy=False)
    B = out["num_atoms"].shape[0]
    n_tot = int(out["num_atoms"].sum())
    if out["frac_x"].shape != (n_tot, 3) or out["atomic_numbers"].shape != (n_tot,) or \
            out["lattice"].shape != (B, 3, 3) or out["idx_start"].shape != (B,):
        raise ValueError("SampleResult arrays do not have the crystals.h5 layout")
    return out


def _is_h5(filename):
    return str(filename).endswith((".h5", ".hdf5"))


def save_sample_results_to_hdf5(crystals: SampleResult, filename: str):
    """process_generated_crystals.py:8-15.  `.h5`/`.hdf5` needs h5py (raises ImportError otherwise: the caller
    asked for HDF5 explicitly); any other name is written as `.npz` with the same keys."""
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    fields = _fields(crystals)
    if _is_h5(filename):
        import h5py
        with h5py.File(filename, "w") as fh:
            group = fh.create_group("crystals")
            for k, v in fields.items():
                group.create_dataset(k, data=v)
    else:
        np.savez(filename, **fields)
    return filename


def load_sample_results_from_hdf5(filename: str) -> SampleResult:
    """process_generated_crystals.py:18-30 (the path is taken as given, not relative to the package)."""
    if _is_h5(filename):
        import h5py
        with h5py.File(filename, "r") as fh:
            data = {k: fh["crystals"][k][:] for k in KEYS}
    else:
        with np.load(filename) as z:
            data = {k: z[k] for k in KEYS}
    return SampleResult(**data)


def get_crystal_indexes(sample_result: SampleResult, sample_idx: int):
    """process_generated_crystals.py:33-37."""
    crystal_start_idx = sample_result.idx_start[sample_idx]
    num_atoms = sample_result.num_atoms[sample_idx]
    return crystal_start_idx, crystal_start_idx + num_atoms


def get_one_crystal(sample_result: SampleResult, sample_idx: int):
    """process_generated_crystals.py:40-47: (lattice [3,3], frac_x [n,3], atomic_numbers [n])."""
    lattice = sample_result.lattice[sample_idx]
    start, end = get_crystal_indexes(sample_result, sample_idx)
    return lattice, sample_result.frac_x[start:end], sample_result.atomic_numbers[start:end]
