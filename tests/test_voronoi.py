"""Voronoi neighbour construction (voronoi_neighbor.py:11-61 restated on scipy / qhull).

pymatgen is not available, so the pins are geometric known answers: lattices whose Voronoi cells are classical polyhedra
(cube, rhombic dodecahedron, truncated octahedron), the 4 pi closure of the solid angles of any bounded cell, symmetry, and
the reference's three filters."""
import numpy as np
import pytest

import scann_oracle as so  # noqa: F401  (conftest puts the package on the path)
from scann.utils import voronoi_neighbor as vn


def test_solid_angle_of_a_cube_face_and_an_octant():
    c = np.zeros(3)
    face = [[1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1]]          # one face of the cube [-1,1]^3 seen from its centre
    assert abs(vn.solid_angle(c, face) - 4 * np.pi / 6) < 1e-12
    tri = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]                           # a triangle spanning one octant
    assert abs(vn.solid_angle(c, tri) - 4 * np.pi / 8) < 1e-12


@pytest.mark.parametrize("name,lattice,n_first,omega_first", [
    ("simple_cubic", 3.0 * np.eye(3), 6, 4 * np.pi / 6),
    ("fcc", 0.5 * 4.0 * np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], float), 12, 4 * np.pi / 12),
])
def test_bravais_lattices_give_the_classical_cells(name, lattice, n_first, omega_first):
    s = vn.Structure(lattice, ["Cu"], [[0, 0, 0]])
    facets = vn.voronoi_polyhedron(lattice, s.cart_coords, 0, 7.0)
    big = [f for f in facets if f[1] > 1e-6]
    assert len(big) == n_first
    assert np.allclose([f[1] for f in big], omega_first, atol=1e-9)
    assert np.allclose(sum(f[1] for f in facets), 4 * np.pi, atol=1e-9)
    d1 = min(f[2] for f in big)
    nb = vn.compute_voronoi_neighbor(s, d_thresh=d1 + 0.01, w_thresh=0.4)
    assert len(nb) == 1 and len(nb[0]) == n_first
    assert all(e[0] == "Cu" and e[1] == 0 and abs(e[3] - 1.0) < 1e-9 and abs(e[4] - d1) < 1e-9 for e in nb[0])


def test_bcc_truncated_octahedron_and_the_ratio_filter():
    a = 3.0
    s = vn.Structure(a * np.eye(3), ["Fe", "Fe"], [[0, 0, 0], [a / 2, a / 2, a / 2]])
    facets = vn.voronoi_polyhedron(s.lattice, s.cart_coords, 0, 7.0)
    hexa = sorted(f[1] for f in facets if abs(f[2] - a * np.sqrt(3) / 2) < 1e-9)
    sq = sorted(f[1] for f in facets if abs(f[2] - a) < 1e-9)
    assert len(hexa) == 8 and len(sq) == 6 and np.allclose(hexa, hexa[0]) and np.allclose(sq, sq[0])
    assert abs(8 * hexa[0] + 6 * sq[0] - 4 * np.pi) < 1e-9 and hexa[0] > sq[0]
    # the reference's filters: weight >= w_thresh, weight / max >= 0.2, distance <= d_thresh (voronoi_neighbor.py:48-50)
    both = vn.compute_voronoi_neighbor(s, d_thresh=4.0, w_thresh=0.4)
    assert [len(x) for x in both] == [14, 14]
    near = vn.compute_voronoi_neighbor(s, d_thresh=2.7, w_thresh=0.4)                 # 8 nearest only
    assert [len(x) for x in near] == [8, 8] and {e[1] for e in near[0]} == {1}
    strong = vn.compute_voronoi_neighbor(s, d_thresh=4.0, w_thresh=sq[0] + 1e-6)      # square faces fall below the weight cut
    assert [len(x) for x in strong] == [8, 8]
    assert all(abs(e[3] - e[2] / hexa[0]) < 1e-12 for e in both[0])


def _benzene():
    ang = np.arange(6) * np.pi / 3
    c = np.stack([1.39 * np.cos(ang), 1.39 * np.sin(ang), np.zeros(6)], 1)
    h = np.stack([2.48 * np.cos(ang), 2.48 * np.sin(ang), np.zeros(6)], 1)
    return vn.Molecule(["C"] * 6 + ["H"] * 6, np.concatenate([c, h]))


def test_boxed_molecule_neighbours_are_symmetric_and_feed_the_model_inputs():
    from scann import _hip
    from scann.utils.general import prepare_input_pmt

    mol = _benzene()
    s = vn.boxed(mol)
    assert np.allclose(np.diag(s.lattice), 10.0) and len(s) == 12
    nb = vn.compute_voronoi_neighbor(s, d_thresh=4.0, w_thresh=0.4)
    assert len(nb) == 12 and all(len(x) >= 3 for x in nb[:6])
    for i in range(6):  # every carbon sees its two ring neighbours and its hydrogen, at the right distances
        got = {e[1]: e for e in nb[i]}
        for j, d in (((i + 1) % 6, 1.39), ((i - 1) % 6, 1.39), (i + 6, 1.09)):
            assert j in got and abs(got[j][4] - d) < 1e-6 and got[j][0] == ("C" if j < 6 else "H")
    # the cubic box keeps the inversion symmetry of the molecule (not its six-fold axis: the outer cells reach the periodic
    # images): atoms i and i + 3 have the same sorted (weight, distance) lists
    key = lambda lst: sorted((round(e[2], 6), round(e[4], 6)) for e in lst)  # noqa: E731
    assert all(key(nb[i]) == key(nb[(i + 3) % 6]) for i in range(6)) and all(key(nb[6 + i]) == key(nb[6 + (i + 3) % 6]) for i in range(6))
    inputs = prepare_input_pmt(s, d_t=4.0, w_t=0.4, angle=True)
    assert inputs["atomic"].tolist() == [[6] * 6 + [1] * 6] and inputs["neighbors"].shape[:2] == (1, 12)
    pk = _hip.pack_inputs(inputs)  # the dict is a valid model input
    assert pk.n_atom == 12 and pk.n_edge == int(inputs["neighbor_mask"].sum()) == sum(len(x) for x in nb)
    ratio = prepare_input_pmt(s, angle=False)["neighbor_weight"]
    assert ratio.max() <= 1.0 + 1e-6 and np.isclose(ratio.max(), 1.0)


def test_dataset_records_xyz_and_the_parallel_driver(tmp_path):
    mol = _benzene()
    rec = {"Atoms": mol.species, "Coords": mol.cart_coords.tolist()}
    crystal = {"Atoms": ["Na", "Cl"], "Coords": [[0, 0, 0], [0.5, 0.5, 0.5]], "Lattice": (5.64 * np.eye(3)).tolist(), "Cartesian": False}
    a = vn.compute_voronoi_neighbor_wrapper(rec, 4.0, 0.4)
    b = vn.compute_voronoi_neighbor_wrapper(crystal, 5.0, 0.4)
    assert len(a) == 12 and len(b) == 2 and {e[0] for e in b[0]} == {"Cl"} and len(b[0]) == 8  # CsCl-type cell: 8 unlike neighbours
    ds = np.empty(2, dtype=object)
    ds[0], ds[1] = rec, crystal
    np.save(tmp_path / "data_energy.npy", ds, allow_pickle=True)
    vn.parallel_compute_neighbor(str(tmp_path / "data_energy.npy"), str(tmp_path / "nei.npy"), d_t=4.0, w_t=0.4, pool=2)
    out = np.load(tmp_path / "nei.npy", allow_pickle=True)
    assert len(out) == 2 and len(out[0]) == 12 and [e[1] for e in out[0][0]] == [e[1] for e in a[0]]
    xyz = tmp_path / "m.xyz"
    xyz.write_text("12\nbenzene\n" + "\n".join("%s %.6f %.6f %.6f" % (s_, *c) for s_, c in zip(mol.species, mol.cart_coords)) + "\n")
    m2 = vn.read_xyz(str(xyz))
    assert isinstance(m2, vn.Molecule) and np.allclose(m2.cart_coords, mol.cart_coords, atol=1e-6)
    ext = tmp_path / "c.xyz"
    ext.write_text('2\nLattice="5.64 0 0 0 5.64 0 0 0 5.64"\nNa 0 0 0\nCl 2.82 2.82 2.82\n')
    c2 = vn.read_xyz(str(ext))
    assert isinstance(c2, vn.Structure) and len(vn.compute_voronoi_neighbor(c2, d_thresh=5.0)[0]) == 8


def test_pathological_unbounded_cells_keep_their_bounded_facets():
    """``allow_pathological=True`` (voronoi_neighbor.py:27): an atom whose Voronoi cell inside the cut-off sphere is UNBOUNDED keeps
    the facets that are bounded and loses the others, no error.  A methane-like cluster alone in a 40 A box (no periodic image within
    the 7 A cut-off): the centre's cell is a regular tetrahedron (four facets of pi each); every outer atom's cell runs to infinity,
    its only bounded facet is the one it shares with the centre -- the same triangle seen from the mirror point, pi again."""
    from scann.utils.voronoi_neighbor import Structure, compute_voronoi_neighbor, voronoi_polyhedron

    t = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=np.float64) / np.sqrt(3.0)
    coords = np.vstack([np.zeros(3), t]) + 20.0
    s = Structure(np.eye(3) * 40.0, ["C", "H", "H", "H", "H"], coords)
    facets = voronoi_polyhedron(s.lattice, s.cart_coords, 0, 7.0)
    assert sorted(j for j, _, _ in facets) == [1, 2, 3, 4] and np.allclose([w for _, w, _ in facets], np.pi, atol=1e-9)
    for i in range(1, 5):
        f = voronoi_polyhedron(s.lattice, s.cart_coords, i, 7.0)
        assert [j for j, _, _ in f] == [0], (i, f)  # the H-H facets have a vertex at infinity: skipped
        assert abs(f[0][1] - np.pi) < 1e-9 and abs(f[0][2] - 1.0) < 1e-12
    nb = compute_voronoi_neighbor(s, cutoff=7, d_thresh=4.0, w_thresh=0.4)
    assert len(nb) == 5 and [len(a) for a in nb] == [4, 1, 1, 1, 1]
    assert all(e[0] == "H" and abs(e[3] - 1.0) < 1e-9 for e in nb[0]) and all(a[0][0] == "C" and a[0][1] == 0 for a in nb[1:])


def test_cutoff_widening_retry_fires_and_gives_up_like_the_reference(capsys):
    """voronoi_neighbor.py:33-60: a tessellation that fails (here: fewer than five sites inside the cut-off sphere) widens the
    cut-off by 5 A and tries again -- the widened value STAYS for the following atoms, as in the reference -- and beyond
    ``max_cutoff`` the atom is given up: its entry is missing from the result (the reference's ``break`` does the same)."""
    from scann.utils.voronoi_neighbor import Structure, compute_voronoi_neighbor

    # simple cubic, a = 3: inside 2.5 A there is only the atom itself; at 7.5 A the classical cell (6 facets of 2 pi / 3)
    s = Structure(np.eye(3) * 3.0, ["Po"], [[0.0, 0.0, 0.0]])
    out = compute_voronoi_neighbor(s, cutoff=2.5, d_thresh=4.0, w_thresh=0.4)
    printed = capsys.readouterr().out
    assert printed.count("Error Voronoi, increase cutoff to") == 1 and "7.5" in printed
    ref = compute_voronoi_neighbor(s, cutoff=7.5, d_thresh=4.0, w_thresh=0.4)
    assert capsys.readouterr().out == ""
    assert len(out) == 1 and len(out[0]) == 6 and sorted(map(tuple, out[0])) == sorted(map(tuple, ref[0]))
    assert np.allclose([e[2] for e in out[0]], 2.0 * np.pi / 3.0, atol=1e-9) and np.allclose([e[4] for e in out[0]], 3.0)
    # one atom alone in a 200 A box: never five sites -- 7, 12, ..., 32 > max_cutoff: given up, no entry, no exception
    lone = Structure(np.eye(3) * 200.0, ["He"], [[100.0, 100.0, 100.0]])
    out = compute_voronoi_neighbor(lone, cutoff=7, d_thresh=4.0, w_thresh=0.4)
    printed = capsys.readouterr().out
    assert out == [] and printed.count("increase cutoff") == 5 and "32.0" in printed and "Error Voronoi, max cutoff" in printed
