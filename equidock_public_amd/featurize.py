"""Graph construction / featurisation of the drop-in: PDB residues -> the k-NN residue graph,
its 27 edge features and the surface feature `mu_r_norm`, i.e. what the reference computes with O(N^2) Python loops in
src/utils/protein_utils.py:201-416 (`protein_to_graph_unbound_bound_residuesonly`, `compute_dig_kNN_graph`).

The O(N^2 * atoms^2) mean all-atom distance matrix, the neighbour selection, the edge features and `mu_r_norm` run on the
MI355X (csrc/eqd_data_kernels.hip: eqd_protein_graph_*), in fp64 like the reference's numpy code (its float32 inputs become
float64 in scipy's cdist and after the float64 Kabsch alignment), so graph indexing is bit-exact - the neighbour set and
order of every residue, int32 endpoints - and features agree to float32 rounding.  What stays on the host is O(N): parsing,
the per-residue local frames and the 3x3 Kabsch alignment of unbound to bound coordinates.

    lig = read_pdb_residues('1AVX_l_b.pdb');  rec = read_pdb_residues('1AVX_r_b.pdb')
    lig, rec, bound_lig_ca, bound_rec_ca, pocket = preprocess_unbound_bound(lig, rec)                 # protein_utils.py:107-175
    lig_g, rec_g = protein_to_graph_unbound_bound(lig, rec, bound_lig_ca, bound_rec_ca, cutoff=30., max_neighbor=10)
    batch = graph.batch_pairs([(dict(lig_g, new_x=lig_g['x']), rec_g)])                               # inference_rigid.py:186-191

No CPU fallback for the device part: without the HIP library the calls raise.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

# residue name -> 21-way embedding id (src/utils/protein_featurizers.py:25-50: residue_type_one_hot_dips_not_one_hot;
# index into its `allowable_set`, 20 = unknown)
_THREE_TO_ONE = {'ALA': 'A', 'ARG': 'R', 'ASN': 'N', 'ASP': 'D', 'CYS': 'C', 'GLN': 'Q', 'GLU': 'E', 'GLY': 'G', 'HIS': 'H',
                 'ILE': 'I', 'LEU': 'L', 'LYS': 'K', 'MET': 'M', 'PHE': 'F', 'PRO': 'P', 'SER': 'S', 'THR': 'T', 'TRP': 'W',
                 'TYR': 'Y', 'VAL': 'V', 'HIP': 'H', 'HIE': 'H', 'TPO': 'T', 'HID': 'H', 'LEV': 'L', 'MEU': 'M', 'PTR': 'Y',
                 'GLV': 'E', 'CYT': 'C', 'SEP': 'S', 'HIZ': 'H', 'CYM': 'C', 'GLM': 'E', 'ASQ': 'D', 'TYS': 'Y', 'CYX': 'C',
                 'GLZ': 'G'}
_ALLOWABLE = ['Y', 'R', 'F', 'G', 'I', 'V', 'A', 'W', 'E', 'H', 'C', 'N', 'M', 'D', 'T', 'S', 'K', 'L', 'Q', 'P']


def residue_type_id(resname):
    one = _THREE_TO_ONE.get(resname)
    return _ALLOWABLE.index(one) if one in _ALLOWABLE else len(_ALLOWABLE)


class Residue:
    """One residue: what the reference carries as a (key, DataFrame) group (src/utils/db5_data.py:15-20)."""
    __slots__ = ('chain', 'number', 'resname', 'atom_names', 'elements', 'coords')

    def __init__(self, chain, number, resname, atom_names, elements, coords):
        self.chain, self.number, self.resname = chain, number, resname
        self.atom_names, self.elements = atom_names, elements
        self.coords = np.asarray(coords, dtype=np.float32).reshape(-1, 3)

    def atom(self, name):
        idx = [i for i, a in enumerate(self.atom_names) if a == name]
        return idx


def read_pdb_residues(path):
    """ATOM records of a PDB file grouped like the reference's `get_residues_db5` (biopandas ATOM table,
    groupby(['chain', 'residue', 'resname']), src/utils/db5_data.py:15-20): groups are ordered by their SORTED keys, not
    by file order, and residues that only differ in their insertion code fall into one group (which the N/CA/C filter
    then drops, protein_utils.py:109-121)."""
    groups = {}
    with open(path) as f:
        for line in f:
            if not line.startswith('ATOM'):
                continue
            name = line[12:16].strip()
            resname = line[17:20].strip()
            chain = line[21:22].strip()
            number = int(line[22:26])
            xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            element = line[76:78].strip() if len(line) >= 78 else ''
            groups.setdefault((chain, number, resname), []).append((name, element, xyz))
    out = []
    for key in sorted(groups):
        atoms = groups[key]
        out.append(Residue(key[0], key[1], key[2], [a[0] for a in atoms], [a[1] for a in atoms], [a[2] for a in atoms]))
    return out


def filter_residues(residues):
    """protein_utils.py:109-121: keep residues with exactly one N, one CA and one C atom."""
    return [r for r in residues if len(r.atom('N')) == 1 and len(r.atom('CA')) == 1 and len(r.atom('C')) == 1]


def alpha_carbon_array(residues):
    """protein_utils.py:139-151 (get_alphaC_loc_array)."""
    locs = [r.coords[r.atom('CA')[0]] for r in residues]
    if len(locs) <= 1:
        locs.append(np.zeros(3, dtype=np.float32))
    return np.stack(locs, axis=0)


def preprocess_unbound_bound(bound_ligand_residues, bound_receptor_residues, graph_nodes='residues', pos_cutoff=8.0,
                             inference=False):
    """protein_utils.py:107-175 (same signature: the reference's callers pass graph_nodes by keyword,
    src/inference_rigid.py:166-168) for graph_nodes == 'residues' (unbound == bound structures, as in the reference):
    filtered residue lists, the bound C-alpha arrays and - unless `inference` - the pocket coordinates (midpoints of
    ligand / receptor C-alpha pairs closer than pos_cutoff; None when there are at most 3)."""
    if graph_nodes != 'residues':
        raise ValueError(f"graph_nodes={graph_nodes!r}: only 'residues' exists (the reference asserts the same, "
                         "src/utils/protein_utils.py:119)")
    lig, rec = filter_residues(bound_ligand_residues), filter_residues(bound_receptor_residues)
    rec_ca, lig_ca = alpha_carbon_array(rec), alpha_carbon_array(lig)
    if inference:
        return lig, rec, lig_ca, rec_ca
    d = np.sqrt(((lig_ca[:, None, :].astype(np.float64) - rec_ca[None, :, :].astype(np.float64)) ** 2).sum(-1))
    al, ar = np.where(d < pos_cutoff)
    pocket = None if al.size <= 3 else 0.5 * (lig_ca[al, :] + rec_ca[ar, :])
    return lig, rec, lig_ca, rec_ca, pocket


def rigid_transform_kabsch_3d(A, B):
    """protein_utils.py:31-64: R (3, 3), t (3, 1) with R A + t ~ B for 3 x N point sets (numpy float64 SVD)."""
    ca, cb = np.mean(A, axis=1, keepdims=True), np.mean(B, axis=1, keepdims=True)
    H = (A - ca) @ (B - cb).T
    U, S, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        R = (Vt.T @ np.diag([1., 1., -1.])) @ U.T
    if abs(np.linalg.det(R) - 1) >= 1e-5:
        raise ValueError("Kabsch: det(R) != 1")
    return R, -R @ ca + cb


def local_frames(residues, residue_loc_is_alphaC=True):
    """protein_utils.py:213-262: per residue the representative location (C-alpha, or the mean of the heavy atoms), and the
    orthonormal frame n_i, u_i, v_i from its N, CA, C atoms - float32 arithmetic like the reference, for all residues at once
    (the reference loops over residues in Python; one residue costs ~35 us that way, the whole protein ~1 ms here)."""
    if len(residues) <= 1:
        raise ValueError("l_or_r contains only 1 residue!")
    counts = np.fromiter((len(r.atom_names) for r in residues), dtype=np.int64, count=len(residues))
    off = np.zeros(len(residues) + 1, dtype=np.int64)
    off[1:] = np.cumsum(counts)
    names = np.asarray([a for r in residues for a in r.atom_names])
    coords = np.concatenate([r.coords for r in residues], 0)
    owner = np.repeat(np.arange(len(residues)), counts)

    def unique_atom(name):
        idx = np.nonzero(names == name)[0]
        if len(idx) != len(residues) or not np.array_equal(owner[idx], np.arange(len(residues))):
            raise ValueError("protein_to_graph: a residue without exactly one N / CA / C atom (filter_residues first)")
        return coords[idx]
    N_loc, ca, C_loc = unique_atom('N'), unique_atom('CA'), unique_atom('C')

    def unit(d):
        return d / np.sqrt((d * d).sum(axis=1, keepdims=True, dtype=np.float32))
    u = unit(N_loc - ca)
    t = unit(C_loc - ca)
    n = unit(np.cross(u, t))
    v = np.cross(n, u)
    if residue_loc_is_alphaC:
        loc = ca
    else:
        heavy = np.asarray([e != 'H' for r in residues for e in r.elements])
        sums = np.zeros((len(residues), 3), dtype=np.float64)
        np.add.at(sums, owner[heavy], coords[heavy].astype(np.float64))
        loc = (sums / np.bincount(owner[heavy], minlength=len(residues))[:, None]).astype(np.float32)
    return loc, n.astype(np.float32), u.astype(np.float32), v.astype(np.float32)


def atoms_ragged(residues):
    """(atom coordinates [A, 3] float32 of all residues one after the other, offsets [n + 1] int32)."""
    off = np.zeros(len(residues) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(r.coords) for r in residues])
    return np.concatenate([r.coords for r in residues], 0).astype(np.float32), off


def knn_graph_device(atoms, atom_off, x, n_i, u_i, v_i, cutoff, max_neighbor, device):
    """compute_dig_kNN_graph (protein_utils.py:311-397) on the device.  atoms [A, 3] float32 + atom_off [n + 1]: all-atom
    coordinates per residue; x, n_i, u_i, v_i [n, 3] float64 (aligned representative locations and frames).
    Returns src, dst (int32, destination-major, the reference's neighbour order), he [E, 27] float32,
    mu_r_norm [n, 5] float32 - device tensors."""
    lib = _lib.load_library()
    dev = torch.device(device)
    n = int(len(atom_off) - 1)
    K = int(max_neighbor)
    if K < 1 or K > 64:
        raise ValueError("max_neighbor must be in 1..64")
    f64 = dict(dtype=torch.float64, device=dev)
    a = _lib.require_device(torch.as_tensor(np.ascontiguousarray(atoms, dtype=np.float32)).to(dev), 'atoms')
    off = torch.as_tensor(np.ascontiguousarray(atom_off, dtype=np.int32)).to(dev)
    xs = [torch.as_tensor(np.ascontiguousarray(t, dtype=np.float64)).to(dev) for t in (x, n_i, u_i, v_i)]
    D = torch.empty(n, n, **f64)
    nbr = torch.empty(n, K, dtype=torch.int32, device=dev)
    nbd = torch.empty(n, K, **f64)
    deg = torch.empty(n, dtype=torch.int32, device=dev)
    mu = torch.empty(n, 5, dtype=torch.float32, device=dev)
    st = _lib.stream_ptr(dev)
    with _lib.device_guard(dev):
        _lib.check(lib.eqd_protein_graph_distances(n, _lib.ptr(a), _lib.ptr(off), _lib.ptr(D), st))
        _lib.check(lib.eqd_protein_graph_select(n, K, C.c_double(float(cutoff)), _lib.ptr(D), _lib.ptr(xs[0]), _lib.ptr(nbr),
                                                _lib.ptr(nbd), _lib.ptr(deg), _lib.ptr(mu), st))
    eoff = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    eoff[1:] = torch.cumsum(deg, 0)
    E, dmin = (int(v) for v in torch.stack([eoff[-1], deg.min() if n else eoff[-1]]).tolist())   # the one host sync
    if n and dmin <= 0:
        # a residue without a neighbour under the cutoff: the reference asserts here (protein_utils.py:354,
        # `assert len(valid_src) > 0`), and its mu_r_norm would be 0 / 0
        raise ValueError(f"a residue has no neighbour closer than cutoff={cutoff}: the reference asserts on such graphs "
                         "(src/utils/protein_utils.py:354)")
    src = torch.empty(E, dtype=torch.int32, device=dev)
    dst = torch.empty(E, dtype=torch.int32, device=dev)
    he = torch.empty(E, 27, dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        _lib.check(lib.eqd_protein_graph_edges(n, K, _lib.ptr(eoff), _lib.ptr(nbr), _lib.ptr(nbd), _lib.ptr(xs[0]),
                                               _lib.ptr(xs[1]), _lib.ptr(xs[2]), _lib.ptr(xs[3]), _lib.ptr(src), _lib.ptr(dst),
                                               _lib.ptr(he), st))
    return src, dst, he, mu


def protein_graph(residues, bound_ca, cutoff, max_neighbor, device, residue_loc_is_alphaC=True):
    """One protein of protein_to_graph_unbound_bound_residuesonly (protein_utils.py:201-416): local frames, alignment of
    the (unbound) representatives onto the bound C-alpha array (:285-309), k-NN graph + features on the device.
    Returns a dict with the node / edge data the reference attaches to its DGL graph: x [n, 3] float32, res_feat [n, 1],
    mu_r_norm [n, 5], src / dst int32, he [E, 27]."""
    loc, n_i, u_i, v_i = local_frames(residues, residue_loc_is_alphaC)
    R, t = rigid_transform_kabsch_3d(loc.T, np.asarray(bound_ca).T)
    x = ((R @ loc.T) + t).T                     # float64 from here on, as in the reference
    n_i, u_i, v_i = (R @ n_i.T).T, (R @ u_i.T).T, (R @ v_i.T).T
    atoms, off = atoms_ragged(residues)
    src, dst, he, mu = knn_graph_device(atoms, off, x, n_i, u_i, v_i, cutoff, max_neighbor, device)
    res = np.asarray([[residue_type_id(r.resname)] for r in residues], dtype=np.float32)
    return {'x': torch.as_tensor(x.astype(np.float32)).to(src.device), 'res_feat': torch.as_tensor(res).to(src.device),
            'mu_r_norm': mu, 'src': src, 'dst': dst, 'he': he}


def protein_to_graph_unbound_bound(unbound_ligand_residues, unbound_receptor_residues, bound_ligand_ca, bound_receptor_ca,
                                   graph_nodes='residues', cutoff=20, max_neighbor=None, one_hot=False,
                                   residue_loc_is_alphaC=True, device='cuda'):
    """The reference's entry point (protein_utils.py:179-198) with its argument order; returns (ligand, receptor) dicts for
    graph.batch_pairs / graph.pair_from_arrays (add the ligand's `new_x`)."""
    if graph_nodes != 'residues' or one_hot:
        raise NotImplementedError("graph_nodes='residues' with the 21-way residue id is what the published models use")
    if max_neighbor is None:
        raise ValueError("max_neighbor is required (graph_max_neighbor, src/utils/args.py:47)")
    lig = protein_graph(unbound_ligand_residues, bound_ligand_ca, cutoff, max_neighbor, device, residue_loc_is_alphaC)
    rec = protein_graph(unbound_receptor_residues, bound_receptor_ca, cutoff, max_neighbor, device, residue_loc_is_alphaC)
    return lig, rec
