"""Neighbour construction: Voronoi-tessellation neighbour lists with solid-angle weights.

Counterpart of the reference's ``scann/utils/voronoi_neighbor.py`` (``compute_voronoi_neighbor`` :11-61, the dataset wrapper
:65-90, ``parallel_compute_neighbor`` :93-130), which delegates to pymatgen's ``VoronoiNN(weight="solid_angle", cutoff=7,
allow_pathological=True)``.  pymatgen is a third-party dependency that is absent here; its published algorithm
(``pymatgen.analysis.local_env.VoronoiNN.get_voronoi_polyhedra`` / ``_extract_cell_info`` / ``solid_angle``) is restated on top
of ``scipy.spatial.Voronoi`` -- the same qhull call pymatgen itself makes:

1. all sites (periodic images included) within ``cutoff`` of atom *i*, sorted by distance (atom *i* itself first);
2. one Voronoi tessellation of that point set; every ridge between point 0 and another point is a facet of atom *i*'s cell;
   ridges with a vertex at infinity are skipped (``allow_pathological=True``);
3. solid angle of a facet seen from the atom: fan triangulation, each triangle by the Van Oosterom-Strackee formula;
4. the reference's filters: ``solid_angle >= w_thresh``, ``solid_angle / max(solid_angle) >= 0.2``, ``distance <= d_thresh``.

Output per atom: ``[[species, index, solid_angle, ratio, distance], ...]`` -- the nested-list format ``DataIterator`` and
``PackedDataset`` read (datagenerator.py:69-90).  The order of an atom's neighbours is qhull's ridge order, as in the
reference; the model does not depend on it (masked softmax over neighbours, attention.py:186-212).

Host-side geometry, not part of the accelerated path.  No pymatgen objects are needed: ``Structure`` / ``Molecule`` below
carry what the algorithm reads; objects that quack like pymatgen's (``.lattice.matrix``, ``.cart_coords``, ``.species``)
are accepted too.
"""
from __future__ import annotations

import itertools
from concurrent.futures import ProcessPoolExecutor

import numpy as np

SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb "
           "Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg "
           "Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
_Z = {s: z for z, s in enumerate(SYMBOLS)}


class Structure:
    """Periodic structure: 3x3 ``lattice`` (rows = lattice vectors, Angstrom), ``species`` (symbols or atomic numbers),
    ``coords`` (Cartesian unless ``coords_are_cartesian=False``)."""

    def __init__(self, lattice, species, coords, coords_are_cartesian=True):
        self.lattice = np.asarray(lattice, dtype=np.float64).reshape(3, 3)
        coords = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        self.cart_coords = coords if coords_are_cartesian else coords @ self.lattice
        self.species = [SYMBOLS[s] if isinstance(s, (int, np.integer)) else str(s) for s in species]
        if len(self.species) != len(self.cart_coords):
            raise ValueError("species and coords differ in length")

    def __len__(self):
        return len(self.species)

    @property
    def atomic_numbers(self):
        return tuple(_Z[s] for s in self.species)


class Molecule:
    """Isolated molecule; ``get_boxed_structure`` puts it into an orthorhombic periodic box the way the reference does before
    the Voronoi step (voronoi_neighbor.py:82-88, general.py:192-198)."""

    def __init__(self, species, coords):
        self.species = [SYMBOLS[s] if isinstance(s, (int, np.integer)) else str(s) for s in species]
        self.cart_coords = np.asarray(coords, dtype=np.float64).reshape(-1, 3)

    def __len__(self):
        return len(self.species)

    @property
    def atomic_numbers(self):
        return tuple(_Z[s] for s in self.species)

    def get_boxed_structure(self, a, b, c):
        centre = 0.5 * (self.cart_coords.max(axis=0) + self.cart_coords.min(axis=0))  # any rigid shift: the result is periodic
        return Structure(np.diag([a, b, c]), self.species, self.cart_coords - centre + 0.5 * np.array([a, b, c]))


def boxed(molecule, box=10.0):
    """The reference's box: every edge max(box, extent + 0.1) (voronoi_neighbor.py:83-87)."""
    ext = molecule.cart_coords.max(axis=0) - molecule.cart_coords.min(axis=0) + 0.1
    return molecule.get_boxed_structure(*(max(box, float(e)) for e in ext))


def _as_arrays(struct):
    """(lattice 3x3, cart coords [n,3], species strings) of a Structure here or a pymatgen-like object."""
    lat = getattr(struct.lattice, "matrix", struct.lattice)
    species = getattr(struct, "species", None)
    species = [getattr(s, "symbol", str(s)) for s in species]
    return np.asarray(lat, dtype=np.float64), np.asarray(struct.cart_coords, dtype=np.float64), species


def sites_in_sphere(lattice, coords, centre, r):
    """All periodic images of ``coords`` within ``r`` of ``centre``: (image coords [m,3], site index [m], distance [m]),
    sorted by distance (pymatgen ``get_sites_in_sphere`` + the sort in ``get_voronoi_polyhedra``)."""
    inv = np.linalg.inv(lattice)
    # images needed along each lattice direction: r / (spacing of the lattice planes), plus the cell the sites sit in
    heights = 1.0 / np.linalg.norm(inv, axis=0)
    nmax = np.ceil(r / heights).astype(int) + 1
    frac = (coords - centre) @ inv
    frac -= np.floor(frac)  # images nearest to the centre's cell
    shifts = np.array(list(itertools.product(*(range(-n, n + 1) for n in nmax))), dtype=np.float64)
    imgs = (frac[None, :, :] + shifts[:, None, :]).reshape(-1, 3) @ lattice
    idx = np.tile(np.arange(len(coords)), len(shifts))
    d = np.linalg.norm(imgs, axis=1)
    keep = d <= r
    order = np.argsort(d[keep], kind="stable")
    return imgs[keep][order] + centre, idx[keep][order], d[keep][order]


def solid_angle(centre, facet):
    """Solid angle of a planar polygon (vertices in order) seen from ``centre``: fan triangulation, every triangle by
    tan(omega / 2) = |r0 . (ri x rj)| / (|r0||ri||rj| + |rj| r0.ri + |ri| r0.rj + |r0| ri.rj)."""
    r = np.asarray(facet, dtype=np.float64) - centre
    rn = np.linalg.norm(r, axis=1)
    angle = 0.0
    for i in range(1, len(r) - 1):
        j = i + 1
        tp = abs(float(np.dot(r[0], np.cross(r[i], r[j]))))
        de = rn[0] * rn[i] * rn[j] + rn[j] * np.dot(r[0], r[i]) + rn[i] * np.dot(r[0], r[j]) + rn[0] * np.dot(r[i], r[j])
        if de == 0:
            a = 0.5 * np.pi if tp > 0 else -0.5 * np.pi
        else:
            a = np.arctan(tp / de)
        angle += (a if a > 0 else a + np.pi) * 2
    return float(angle)


def voronoi_polyhedron(lattice, coords, i, cutoff):
    """Facets of atom ``i``'s Voronoi cell: list of (site index, solid angle, distance), in qhull's ridge order.
    Raises RuntimeError when the tessellation fails (too few points in the cutoff sphere)."""
    from scipy.spatial import Voronoi
    from scipy.spatial import QhullError

    pts, idx, dist = sites_in_sphere(lattice, coords, coords[i], cutoff)
    if len(pts) < 5:
        raise RuntimeError("too few sites within the cutoff")
    try:
        vor = Voronoi(pts)
    except QhullError as e:
        raise RuntimeError(str(e))
    out = []
    for (p, q), vind in vor.ridge_dict.items():
        if p != 0 and q != 0:
            continue
        other = q if p == 0 else p
        if -1 in vind:
            continue  # facet with a vertex at infinity (allow_pathological=True)
        out.append((int(idx[other]), solid_angle(pts[0], vor.vertices[vind]), float(dist[other])))
    return out


def compute_voronoi_neighbor(struct, cutoff=7, d_thresh=4.0, w_thresh=0.4, max_cutoff=30):
    """Reference signature and output (voronoi_neighbor.py:11-61): per atom ``[[species, index, solid_angle, ratio, distance]]``
    of the facets with solid_angle >= w_thresh, ratio = solid_angle / max >= 0.2 and distance <= d_thresh; a failing
    tessellation widens the cutoff in steps of 5 A up to ``max_cutoff`` (the reference's retry loop, :33-60)."""
    lattice, coords, species = _as_arrays(struct)
    local = []
    for i in range(len(coords)):
        while True:
            try:
                nns = voronoi_polyhedron(lattice, coords, i, cutoff)
                if not nns:
                    raise RuntimeError("no bounded facet")
                wmax = max(w for _, w, _ in nns)
                local.append([[species[j], j, w, w / wmax, d] for j, w, d in nns if w >= w_thresh and w / wmax >= 0.2 and d <= d_thresh])
                break
            except RuntimeError:
                cutoff += 5.0
                print("Error Voronoi, increase cutoff to ", cutoff)
                if cutoff > max_cutoff:
                    print("Error Voronoi, max cutoff")
                    break
    return local


def structure_from_record(s, box=10):
    """Dataset record ``{'Atoms', 'Coords', ['Lattice', 'Cartesian']}`` (written by the reference's dataset builders) ->
    periodic Structure; molecules are boxed (voronoi_neighbor.py:65-88)."""
    coords = np.array(s["Coords"], dtype="float32")
    if "Lattice" in s:
        return Structure(s["Lattice"], s["Atoms"], coords, coords_are_cartesian=s["Cartesian"] if "Cartesian" in s else True)
    return boxed(Molecule(s["Atoms"], coords), box)


def compute_voronoi_neighbor_wrapper(s, d_t, w_t, box=10):
    return compute_voronoi_neighbor(structure_from_record(s, box), 7, d_t, w_t)


def parallel_compute_neighbor(dataset_path, save_path, d_t=4.0, w_t=0.2, pool=8):
    """Neighbour lists of a whole ``*_data_energy.npy`` dataset, ``pool`` processes (voronoi_neighbor.py:93-130)."""
    dataset = np.load(dataset_path, allow_pickle=True)
    print("Computing Voronoi neighbor for dataset ", dataset_path, ", parallel process: ", pool, ", saving to: ", save_path)
    all_data = []
    with ProcessPoolExecutor(pool) as executor:
        for i in range(0, len(dataset), pool):
            if i % (10 * pool) == 0:
                print(i)
            futures = [executor.submit(compute_voronoi_neighbor_wrapper, s, d_t, w_t) for s in dataset[i:i + pool]]
            all_data.extend(f.result() for f in futures)
    print("Saving data")
    out = np.empty(len(all_data), dtype=object)
    for i, a in enumerate(all_data):
        out[i] = a
    np.save(save_path, out)
    print("Finished computing Voronoi neighbor for dataset ", dataset_path)


def read_xyz(path):
    """xyz file (optionally extended: ``Lattice="ax ay az bx ..."`` on the comment line) -> Molecule or Structure
    (general.py:147-175, with the reference's 'Latiice' key typo not reproduced)."""
    with open(path) as f:
        lines = f.read().splitlines()
    n = int(lines[0].split()[0])
    lattice = None
    if 'Lattice="' in lines[1]:
        lattice = np.array([float(x) for x in lines[1].split('Lattice="')[1].split('"')[0].split()]).reshape(3, 3)
    atoms, coords = [], []
    for line in lines[2:2 + n]:
        t = line.split()
        atoms.append(t[0])
        coords.append([float(t[1]), float(t[2]), float(t[3])])
    return Structure(lattice, atoms, coords) if lattice is not None else Molecule(atoms, coords)
