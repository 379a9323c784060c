"""ORACLE-ONLY TOOL: tests/golden/graph_case.npz from the REAL reference graph construction, in THIS container.

src/utils/protein_utils.py cannot be imported (it pulls in dgllife through protein_featurizers.py and DGL), so - like
oracle/make_golden_loss.py - the functions are taken out of the reference's own source files with `ast` and executed as
they are, with the oracle's DGL stand-in for the graph container (nothing is copied into this repository: only numbers
leave this script):
    rigid_transform_Kabsch_3D (:31-64), distance_list_featurizer (:71-86), residue_list_featurizer_dips_NOT_one_hot,
    preprocess_unbound_bound (:107-175), protein_to_graph_unbound_bound_residuesonly (:201-416)
    and residue_type_one_hot_dips_not_one_hot (src/utils/protein_featurizers.py:25-50).
Input: a real complex of the reference's DB5.5 copy (data/benchmark5.5/structures/1GL1_{l,r}_b.pdb), parsed by the
product's PDB reader into the (key, DataFrame) groups the reference gets from biopandas (absent here) - the receptor is
cut to its first 110 residues to keep the fixture small.  Recorded: the residues' atoms (inputs) and, per protein, the
reference's src / dst / he / x / mu_r_norm / res_feat, the bound C-alpha arrays and the pocket coordinates.

    python oracle/make_golden_graph.py
"""
import ast
import math
import os
import sys

import numpy as np
import pandas as pd
import scipy.spatial as spa
import torch
from numpy import linalg as LA
from scipy.special import softmax

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, '_dgl_standin'))
import dgl  # noqa: E402  (the stand-in)

from equidock_public_amd import featurize as F  # noqa: E402

REF = '/root/reference/src/utils/'
PDB = '/root/reference/data/benchmark5.5/structures/1GL1_%s_b.pdb'
CUTOFF, MAX_NEIGHBOR = 30.0, 10          # src/utils/args.py:46-47 (graph_cutoff, graph_max_neighbor)


def reference_functions():
    ns = {'np': np, 'spa': spa, 'LA': LA, 'dgl': dgl, 'math': math, 'softmax': softmax, 'torch': torch,
          'zerocopy_from_numpy': torch.from_numpy}
    for path, names in ((REF + 'protein_featurizers.py', ['residue_type_one_hot_dips_not_one_hot']),
                        (REF + 'protein_utils.py', ['rigid_transform_Kabsch_3D', 'distance_list_featurizer',
                                                    'residue_list_featurizer_dips_NOT_one_hot', 'preprocess_unbound_bound',
                                                    'protein_to_graph_unbound_bound_residuesonly'])):
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns


def as_groups(residues):
    """product Residue objects -> the (key, DataFrame) groups of df.groupby(['chain', 'residue', 'resname'])"""
    out = []
    for r in residues:
        df = pd.DataFrame({'x': r.coords[:, 0].astype(np.float64), 'y': r.coords[:, 1].astype(np.float64),
                           'z': r.coords[:, 2].astype(np.float64), 'atom_name': r.atom_names, 'element': r.elements,
                           'resname': [r.resname] * len(r.atom_names), 'residue': [r.number] * len(r.atom_names),
                           'chain': [r.chain] * len(r.atom_names)})
        out.append(((r.chain, r.number, r.resname), df))
    return out


def pack_residues(residues, prefix, out):
    atoms, off = F.atoms_ragged(residues)
    out[prefix + 'atoms'], out[prefix + 'atom_off'] = atoms, off
    out[prefix + 'atom_names'] = np.asarray([a for r in residues for a in r.atom_names], dtype='U4')
    out[prefix + 'elements'] = np.asarray([e for r in residues for e in r.elements], dtype='U2')
    out[prefix + 'resnames'] = np.asarray([r.resname for r in residues], dtype='U3')
    out[prefix + 'chains'] = np.asarray([r.chain for r in residues], dtype='U1')
    out[prefix + 'numbers'] = np.asarray([r.number for r in residues], dtype=np.int32)


def main():
    ns = reference_functions()
    lig_all = F.read_pdb_residues(PDB % 'l')
    rec_all = F.read_pdb_residues(PDB % 'r')
    rec_all = rec_all[:112]
    # the reference's own pipeline (src/utils/db5_data.py:111, 146): preprocess, then graphs
    lig_f, rec_f, lig_ca, rec_ca, pocket = ns['preprocess_unbound_bound'](as_groups(lig_all), as_groups(rec_all), 'residues',
                                                                          pos_cutoff=8.0, inference=False)
    gl, gr = ns['protein_to_graph_unbound_bound_residuesonly'](lig_f, rec_f, lig_ca, rec_ca, cutoff=CUTOFF,
                                                               max_neighbor=MAX_NEIGHBOR, one_hot=False,
                                                               residue_loc_is_alphaC=True)
    # the product's host-side restatements must agree before anything is written
    p_lig, p_rec, p_lig_ca, p_rec_ca, p_pocket = F.preprocess_unbound_bound(lig_all, rec_all)
    assert len(p_lig) == len(lig_f) and len(p_rec) == len(rec_f)
    assert np.array_equal(p_lig_ca, lig_ca) and np.array_equal(p_rec_ca, rec_ca)
    assert np.allclose(p_pocket, pocket, atol=0, rtol=0)
    out = {'cutoff': CUTOFF, 'max_neighbor': MAX_NEIGHBOR, 'lig_ca': lig_ca.astype(np.float32),
           'rec_ca': rec_ca.astype(np.float32), 'pocket': pocket.astype(np.float32)}
    pack_residues(lig_all, 'lig_in_', out)
    pack_residues(rec_all, 'rec_in_', out)
    for nm, g in (('lig', gl), ('rec', gr)):
        s, d = g.edges()
        out[nm + '_src'], out[nm + '_dst'] = s.numpy().astype(np.int32), d.numpy().astype(np.int32)
        out[nm + '_he'] = g.edata['he'].numpy()
        out[nm + '_x'] = g.ndata['x'].numpy()
        out[nm + '_mu'] = g.ndata['mu_r_norm'].numpy()
        out[nm + '_res'] = g.ndata['res_feat'].numpy()
        print(nm, 'nodes', g.num_nodes(), 'edges', g.num_edges(), 'he', out[nm + '_he'].shape, out[nm + '_he'].dtype)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'graph_case.npz'), **out)
    print('written', os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'graph_case.npz')), 'bytes')


# Round 3: three more complexes of the reference's DB5.5 copy, each through the reference's own
# function as above.  To keep the fixtures small the edge features are stored for every `he_stride`-th edge only (the
# int32 endpoints - the "bit-exact graph indexing" claim - are complete), and the partner protein that is not the point of
# the case is cut short.  The reference call is timed: that is the CPU baseline of the graph-construction row.
EXTRA = {
    # name: (pdb id, ligand residue cut, receptor residue cut, he stride, what the case pins)
    'big': ('1DE4', None, 40, 8, '1 270-residue ligand (the >= 1 000-residue regime: argsort over 1 269 candidates per row)'),
    'pair300': ('2J7P', None, None, 4, 'a DIPS-sized pair, 271 + 298 residues'),
    'tiny': ('1PPE', 8, 30, 1, 'fewer residues than max_neighbor: every row keeps all its candidates in np.where order'),
}


def main_extra():
    import time
    ns = reference_functions()
    lines = []
    for name, (pid, lcut, rcut, stride, what) in EXTRA.items():
        pdb = '/root/reference/data/benchmark5.5/structures/' + pid + '_%s_b.pdb'
        lig_all, rec_all = F.read_pdb_residues(pdb % 'l'), F.read_pdb_residues(pdb % 'r')
        lig_all = lig_all[:lcut] if lcut else lig_all
        rec_all = rec_all[:rcut] if rcut else rec_all
        lig_f, rec_f, lig_ca, rec_ca = ns['preprocess_unbound_bound'](as_groups(lig_all), as_groups(rec_all), 'residues',
                                                                      pos_cutoff=8.0, inference=True)
        t0 = time.perf_counter()
        gl, gr = ns['protein_to_graph_unbound_bound_residuesonly'](lig_f, rec_f, lig_ca, rec_ca, cutoff=CUTOFF,
                                                                   max_neighbor=MAX_NEIGHBOR, one_hot=False,
                                                                   residue_loc_is_alphaC=True)
        dt = time.perf_counter() - t0
        p = F.preprocess_unbound_bound(lig_all, rec_all, inference=True)
        assert len(p[0]) == len(lig_f) and len(p[1]) == len(rec_f)
        assert np.array_equal(p[2], lig_ca) and np.array_equal(p[3], rec_ca)
        out = {'cutoff': CUTOFF, 'max_neighbor': MAX_NEIGHBOR, 'he_stride': stride, 'lig_ca': lig_ca.astype(np.float32),
               'rec_ca': rec_ca.astype(np.float32), 'what': what, 'pdb': pid, 'reference_seconds': dt}
        pack_residues(lig_all, 'lig_in_', out)
        pack_residues(rec_all, 'rec_in_', out)
        for nm, g in (('lig', gl), ('rec', gr)):
            s, d = g.edges()
            out[nm + '_src'], out[nm + '_dst'] = s.numpy().astype(np.int32), d.numpy().astype(np.int32)
            out[nm + '_he'] = g.edata['he'].numpy()[::stride].copy()
            out[nm + '_x'] = g.ndata['x'].numpy()
            out[nm + '_mu'] = g.ndata['mu_r_norm'].numpy()
            out[nm + '_res'] = g.ndata['res_feat'].numpy()
        path = os.path.join(ROOT, 'tests', 'golden', f'graph_case_{name}.npz')
        np.savez_compressed(path, **out)
        line = (f'{name}: {pid} ligand {gl.num_nodes()} residues / {gl.num_edges()} edges, receptor {gr.num_nodes()} / '
                f'{gr.num_edges()}; reference protein_to_graph_unbound_bound_residuesonly (this container, 1 thread): '
                f'{dt:.2f} s; fixture {os.path.getsize(path)} bytes')
        print(line)
        lines.append(line)
    # reference timing alone at the sizes of the measurement table (no fixture): 200 / 1 000 / 2 000-residue proteins
    for pid, side, cut in (('1N2C', 'r', 200), ('1N2C', 'r', 1000), ('1N2C', 'r', 2000)):
        pdb = '/root/reference/data/benchmark5.5/structures/' + pid + '_%s_b.pdb'
        big = F.read_pdb_residues(pdb % side)[:cut]
        small = F.read_pdb_residues(pdb % 'l')[:20]
        a, b, a_ca, b_ca = ns['preprocess_unbound_bound'](as_groups(big), as_groups(small), 'residues', pos_cutoff=8.0, inference=True)
        t0 = time.perf_counter()
        ns['protein_to_graph_unbound_bound_residuesonly'](a, b, a_ca, b_ca, cutoff=CUTOFF, max_neighbor=MAX_NEIGHBOR,
                                                          one_hot=False, residue_loc_is_alphaC=True)
        line = (f'reference graph construction, {len(a)} + {len(b)} residues ({pid}): {time.perf_counter() - t0:.2f} s '
                f'(this container, 1 thread)')
        print(line)
        lines.append(line)
    with open(os.path.join(ROOT, 'profiles', 'r03_graph_reference_cpu.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    if '--extra' in sys.argv:
        main_extra()
    else:
        main()
