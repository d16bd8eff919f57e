"""Host batch helpers with the reference's names and semantics (scann/utils/general.py:14-246).
The parts that feed the forward path, plus the README's ``load_file`` for the formats that need no pymatgen (xyz, extended
xyz, POSCAR); the other pymatgen / openbabel file loaders are out of scope."""
from __future__ import annotations

import os

import numpy as np


def pad_sequence(sequences, maxlen=None, dtype="int32", value=0, padding="post"):
    """Post-pad a list of variable-length sequences to ``[len(sequences), maxlen, ...]``; sequences
    longer than ``maxlen`` keep their last ``maxlen`` items (general.py:14-32)."""
    if maxlen is None:
        maxlen = max(len(s) for s in sequences)
    trailing = np.asarray(sequences[0]).shape[1:]
    out = np.full((len(sequences), maxlen) + trailing, value, dtype=dtype)
    for i, s in enumerate(sequences):
        s = np.asarray(s[-maxlen:] if maxlen else s[:0], dtype=dtype)
        if len(s):
            out[i, : len(s)] = s
    return out


def pad_nested_sequences(sequences, max_len_1, max_len_2, dtype="int32", value=0):
    """Pad a ragged 3-D list to ``[B, max_len_2, max_len_1]`` (general.py:35-50)."""
    out = np.full((len(sequences), max_len_2, max_len_1), value, dtype=dtype)
    for b, outer in enumerate(sequences):
        outer = outer[-max_len_2:]
        for a, inner in enumerate(outer):
            inner = inner[-max_len_1:]
            if len(inner):
                out[b, a, : len(inner)] = np.asarray(inner, dtype=dtype)
    return out


def split_data(len_data, test_percent=0.1, train_size=None, test_size=None):
    """Random train/valid/test split with the reference's sizing rule (general.py:79-101)."""
    if train_size:
        n_train, n_test = train_size, test_size
    else:
        n_train = int(len_data * (1 - test_percent * 2))
        n_test = int(len_data * test_percent)
    n_val = len_data - n_train - n_test
    perm = np.random.permutation(len_data)
    train, valid, test, extra = np.split(perm, [n_train, n_train + n_val, n_train + n_val + n_test])
    return train, valid, test, extra


def load_dataset(dataset, dataset_neighbor, target_prop, use_ref=False, use_ring=True):
    """Load the ``*_data_energy.npy`` / ``*_data_neighbor*.npy`` object arrays written by the
    reference's preprocessing (general.py:104-144)."""
    data_full = np.load(dataset, allow_pickle=True)
    if use_ref:
        print("Using reference energy optimization", "\n")
    if use_ring:
        print("Using ring aromatic information", "\n")
    rows = []
    for d in data_full:
        if use_ring:
            rows.append([d["Atomic"], float(d["Properties"][target_prop]),
                         np.stack([d["Features"][x] for x in d["Features"]], -1)])
        elif use_ref:
            rows.append([d["Atomic"], float(d["Properties"][target_prop]) - float(d["Properties"]["Ref_energy"])])
        else:
            rows.append([d["Atomic"], float(d["Properties"][target_prop])])
    data_energy = np.array(rows, dtype="object")  # [n, 2 | 3] object array: SCANN.prepare_dataset indexes column 1
    data_neighbor = np.array(np.load(dataset_neighbor, allow_pickle=True), dtype="object")
    return data_energy, data_neighbor


def prepare_input_from_neighbors(atomic_numbers, neighbors, angle=True):
    """Input dict for ONE structure from its per-atom neighbour lists -- the second half of the reference's
    ``prepare_input_pmt`` (general.py:220-244).  ``neighbors[a]`` is the list the Voronoi step produces for atom ``a``:
    ``[species, index, solid_angle, ratio, distance]`` entries (voronoi_neighbor.py:38-47); ``angle`` selects the raw solid
    angle (SCANN+, g_update) or the normalised ratio, as in the reference."""
    local_neighbor = np.array([pad_sequence([[n[1] for n in lc] for lc in neighbors], value=1000)], "int32")
    mask_local = local_neighbor != 1000
    local_neighbor[~mask_local] = 0
    wi = 2 if angle else 3
    local_weight = np.array([pad_sequence([[n[wi] for n in lc] for lc in neighbors], dtype="float32")])
    local_distance = np.array([pad_sequence([[n[-1] for n in lc] for lc in neighbors], dtype="float32")])
    atomics = np.array([list(atomic_numbers)], "int32")
    return {
        "atomic": atomics,
        "atom_mask": np.expand_dims(atomics != 0, -1),
        "neighbors": local_neighbor,
        "neighbor_mask": mask_local,
        "neighbor_weight": local_weight,
        "neighbor_distance": local_distance,
    }


def prepare_input_pmt(struct, d_t=4.0, w_t=0.4, angle=True):
    """README entry point (general.py:206-246): model inputs of ONE structure -- Voronoi neighbour lists (solid-angle weights,
    filtered by ``w_t`` / ``d_t``) padded into the Keras input dict.  ``struct``: a periodic ``Structure`` of
    ``scann.utils.voronoi_neighbor`` (box a ``Molecule`` first: ``boxed(mol)``), or a pymatgen ``Structure`` (anything with
    ``.lattice.matrix``, ``.cart_coords``, ``.species``, ``.atomic_numbers``).  The tessellation itself is
    ``voronoi_neighbor.compute_voronoi_neighbor`` (scipy / qhull; no pymatgen needed)."""
    from .voronoi_neighbor import compute_voronoi_neighbor

    neighbors = compute_voronoi_neighbor(struct, d_thresh=d_t, w_thresh=w_t)
    return prepare_input_from_neighbors(struct.atomic_numbers, neighbors, angle)


def process_xyz_pmt(file):
    """``{"Atoms", "Coords"[, "Lattice"]}`` of an xyz file whose comment line may carry ``Lattice="..."`` (general.py:147-175).
    The reference stores the cell under the misspelt key ``"Latiice"``; both spellings are set here."""
    from .voronoi_neighbor import Structure, read_xyz

    st = read_xyz(file)
    out = {"Atoms": list(st.species), "Coords": [list(map(float, c)) for c in st.cart_coords]}
    if isinstance(st, Structure):
        out["Lattice"] = out["Latiice"] = [list(map(float, row)) for row in st.lattice]
    return out


def _read_poscar(path):
    """VASP 5 POSCAR / CONTCAR: scale, three lattice vectors, element symbols, counts, [Selective dynamics], Direct | Cartesian."""
    from .voronoi_neighbor import Structure

    with open(path) as f:
        lines = [ln.strip() for ln in f.read().splitlines()]
    scale = float(lines[1].split()[0])
    lat = np.array([[float(x) for x in lines[i].split()[:3]] for i in (2, 3, 4)])
    lat = lat * scale if scale > 0 else lat * (abs(scale) / abs(np.linalg.det(lat))) ** (1.0 / 3.0)
    symbols = lines[5].split()
    if symbols[0].isdigit():
        raise ValueError("POSCAR without an element line (VASP 4 format)")
    counts = [int(x) for x in lines[6].split()]
    k = 7
    if lines[k][:1] in "sS":
        k += 1
    direct = lines[k][:1] not in "cCkK"
    n = sum(counts)
    xyz = np.array([[float(x) for x in lines[k + 1 + i].split()[:3]] for i in range(n)])
    species = [s for s, c in zip(symbols, counts) for _ in range(c)]
    if not direct:
        xyz = xyz * (scale if scale > 0 else 1.0)
    return Structure(lat, species, xyz, coords_are_cartesian=not direct)


def load_file(file, mol=False):
    """README entry point (general.py:178-203): the structure in ``file`` as an object ``prepare_input_pmt`` accepts.
    ``mol=True`` (or an xyz file without a cell): the molecule is put into the reference's periodic box, every edge
    max(10, extent + 0.1) Angstrom.  xyz / extended xyz / POSCAR are read here; for anything else pymatgen is used when it is
    installed.  Like the reference, a file that cannot be read yields a message and ``None``."""
    from .voronoi_neighbor import Molecule, Structure, boxed, read_xyz

    try:
        name = os.path.basename(str(file))
        ext = os.path.splitext(name)[1].lower()
        if ext in (".xyz", ".extxyz"):
            st = read_xyz(file)
            if isinstance(st, Molecule) or mol:
                st = boxed(st if isinstance(st, Molecule) else Molecule(st.species, st.cart_coords))
            return st
        if name.upper().startswith(("POSCAR", "CONTCAR")) or ext in (".vasp", ".poscar"):
            st = _read_poscar(file)
            return boxed(Molecule(st.species, st.cart_coords)) if mol else st
        try:
            from pymatgen.core import Molecule as PmgMolecule, Structure as PmgStructure
        except ImportError:
            raise ValueError("format needs pymatgen, which is not installed")
        if mol:
            m = PmgMolecule.from_file(file)
            return boxed(Molecule([str(s) for s in m.species], m.cart_coords))
        return PmgStructure.from_file(file)
    except Exception as e:  # the reference swallows every error here and returns None (general.py:201-203)
        print("Can not read file using Pymatgen. Please check the file format", "(%s)" % e)
        return None
