"""The sampling path's side of the reference's visualisation module (diffusion/inference/visualize_crystal.py:16-20,
57-67): the VisualizationSetting enum and `vis_crystal_during_sampling`, which DiffusionLoss.sample calls on the
reference's schedule (every 10th timestep for ALL, every timestep for ALL_DETAILED, the final state for LAST and up).

The reference renders a plotly figure and writes `<name>.png` (plotly + kaleido + pymatgen, none of which this build
depends on).  What the sampler has to provide at those points is the FRAME -- species, cell and fractional coordinates
of the intermediate state -- so this module writes it as a P1 CIF, `<name>.cif`, which any structure viewer (including
the reference's own plot_crystal via pymatgen) opens.  A batch of several crystals writes `<name>_<b>.cif` per crystal
(the reference's call only works for a batch of one: it squeezes the batch dimension of the lattice)."""
from enum import Enum

import numpy as np

from ..tools.atomic_number_table import AtomicNumberTable, atomic_number_indexes_to_atomic_numbers, symbol_of


class VisualizationSetting(Enum):
    NONE = 0
    LAST = 1
    ALL = 2
    ALL_DETAILED = 3


def _cell_parameters(lattice):
    a, b, c = (np.asarray(lattice, dtype=np.float64)[i] for i in range(3))
    la, lb, lc = (float(np.linalg.norm(v)) for v in (a, b, c))
    ang = lambda u, v, lu, lv: float(np.degrees(np.arccos(np.clip(np.dot(u, v) / max(lu * lv, 1e-300), -1.0, 1.0))))
    return la, lb, lc, ang(b, c, lb, lc), ang(a, c, la, lc), ang(a, b, la, lb)


def write_cif(path, atomic_numbers, lattice, frac_x, title="arreau_amd sample"):
    """Minimal P1 CIF: cell lengths / angles from the row-vector lattice, one atom site per row of frac_x (wrapped
    into [0, 1)).  The mask state (atomic number beyond the periodic table) is written as the dummy species X."""
    la, lb, lc, al, be, ga = _cell_parameters(lattice)
    import os
    lines = [f"data_{os.path.basename(title).replace(' ', '_')}", "_symmetry_space_group_name_H-M 'P 1'", "_symmetry_Int_Tables_number 1",
             f"_cell_length_a {la:.6f}", f"_cell_length_b {lb:.6f}", f"_cell_length_c {lc:.6f}",
             f"_cell_angle_alpha {al:.6f}", f"_cell_angle_beta {be:.6f}", f"_cell_angle_gamma {ga:.6f}",
             "loop_", "_atom_site_label", "_atom_site_type_symbol", "_atom_site_fract_x", "_atom_site_fract_y",
             "_atom_site_fract_z"]
    frac = np.mod(np.asarray(frac_x, dtype=np.float64), 1.0)
    for i, (z, f) in enumerate(zip(np.asarray(atomic_numbers).reshape(-1), frac)):
        sym = symbol_of(int(z))
        lines.append(f"{sym}{i + 1} {sym} {f[0]:.6f} {f[1]:.6f} {f[2]:.6f}")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return path


def vis_crystal_during_sampling(z_table: AtomicNumberTable, A, lattice, frac_x, name: str, show_bonds: bool = False,
                                num_atoms=None):
    """visualize_crystal.py:57-67.  A [N] class indices, lattice [B,3,3] (or [3,3]), frac_x [N,3]; num_atoms [B] splits a
    batch.  Returns the list of files written.  `show_bonds` only affects the reference's rendering."""
    A = np.asarray(A).reshape(-1)
    lattice = np.asarray(lattice, dtype=np.float64).reshape(-1, 3, 3)
    frac_x = np.asarray(frac_x, dtype=np.float64).reshape(-1, 3)
    zs = atomic_number_indexes_to_atomic_numbers(z_table, A)
    B = lattice.shape[0]
    counts = [len(A)] if num_atoms is None else [int(v) for v in np.asarray(num_atoms).reshape(-1)]
    if len(counts) != B or sum(counts) != len(A):
        raise ValueError("num_atoms must hold one count per lattice and add up to the number of atoms")
    out, first = [], 0
    for b, n in enumerate(counts):
        path = f"{name}.cif" if B == 1 else f"{name}_{b}.cif"
        out.append(write_cif(path, zs[first:first + n], lattice[b], frac_x[first:first + n], title=name))
        first += n
    return out
