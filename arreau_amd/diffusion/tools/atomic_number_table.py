"""Class-index <-> atomic-number table (host logic).

Mirrors diffusion/tools/atomic_number_table.py:7-50 of the reference (same class and function
names, so a pickled `AtomicNumberTable` inside a checkpoint's hyper_parameters can be remapped
onto this class).  The mask state is atomic number 2001 and is always the last class after sorting.
"""
from typing import Sequence

import numpy as np


class AtomicNumberTable:
    MASK_ATOMIC_NUMBER = 2001

    def __init__(self, zs: Sequence[int]):
        self.zs = list(zs)

    def __len__(self) -> int:
        return len(self.zs)

    def __str__(self):
        return f"AtomicNumberTable: {tuple(s for s in self.zs)}"

    def index_to_z(self, index: int) -> int:
        return self.zs[index]

    def z_to_index(self, atomic_number: int) -> int:
        return self.zs.index(atomic_number)


def get_atomic_number_table_from_zs(zs) -> AtomicNumberTable:
    z_set = set()
    for group in zs:
        z_set.update(int(z) for z in group)
    z_set.add(AtomicNumberTable.MASK_ATOMIC_NUMBER)
    return AtomicNumberTable(sorted(z_set))


def atomic_number_indexes_to_atomic_numbers(z_table: AtomicNumberTable, atomic_number_indexes) -> np.ndarray:
    lut = np.asarray(z_table.zs, dtype=np.int64)
    return lut[np.asarray(atomic_number_indexes, dtype=np.int64)]


def atomic_numbers_to_indices(z_table: AtomicNumberTable, atomic_numbers) -> np.ndarray:
    return np.asarray([z_table.z_to_index(int(z)) for z in np.asarray(atomic_numbers).reshape(-1)], dtype=np.int64)


# symbol -> Z without pymatgen (reference: atomic_symbols_to_indices, atomic_number_table.py:84-89)
_SYMBOLS = (
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
    "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir "
    "Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl "
    "Mc Lv Ts Og").split()
SYMBOL_TO_Z = {s: i + 1 for i, s in enumerate(_SYMBOLS)}


def symbol_of(atomic_number: int) -> str:
    """Element symbol; "X" (the CIF dummy species) for the mask state / anything beyond the table."""
    return _SYMBOLS[atomic_number - 1] if 1 <= atomic_number <= len(_SYMBOLS) else "X"


def atomic_symbols_to_indices(z_table: AtomicNumberTable, atomic_symbols) -> np.ndarray:
    return atomic_numbers_to_indices(z_table, [SYMBOL_TO_Z[s] for s in atomic_symbols])
