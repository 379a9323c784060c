"""Batched ligand/receptor pair-graph container: the input boundary of the hot path.

The reference hands its model a batched DGL heterograph built by
`hetero_graph_from_sg_l_r_pair` + `dgl.batch` (src/utils/train_utils.py:61-100).  DGL is a
third-party pin that is absent here, so this module provides the accessor subset the callers
and the model use:

    g.nodes['ligand'].data['new_x'], g.nodes['receptor'].data['x'], ...['res_feat'], ['mu_r_norm']
    g.edges['ll'].data['he'], g.edges['rr'].data['he']
    g.batch_num_nodes('ligand'), g.to(device), batch(list_of_pairs), unbatch(g)

and, for the HIP path, `PairBatch.pack()` which lays the batch out for the kernels:

  * ONE node array: all ligand nodes of all pairs, then all receptor nodes (ll and rr edges
    share every weight, rigid_docking_model.py:236-237,263-265, so one launch covers both);
  * edges destination-sorted (they already are: protein_utils.py:339-346; validated here,
    stably sorted otherwise) -> CSR `rowptr` by destination, CSC (`csc_ptr`, `csc_eid`) by
    source for the atomics-free backward scatter;
  * node-aligned edge tiles of <= 32 edges (`tile_node`) so the per-destination mean
    (DGL copy_edge + mean, rigid_docking_model.py:274-283) is a within-tile reduction;
  * per-pair segment offsets + a work list of 32-node blocks for the block-diagonal
    cross-attention (replaces the dense get_mask, rigid_docking_model.py:68-78).

All index arrays are int32 (graph indexing is bit-exact with the reference's int32 edges).
"""
import ctypes as _C
import os

import numpy as np
import torch

TILE_EDGES = 32      # must equal eqd_tile_edges() of the C-ABI library
TILE_NODES = 32
ATT_BLOCK = 32

# ---- native host pack (csrc_host/eqd_host_pack.cpp -> libequidock_host.so; plain C++, safe in DataLoader workers) ----

class _HostPackIn(_C.Structure):
    _fields_ = [('n_pairs', _C.c_int32), ('n_lig', _C.c_int32), ('n_rec', _C.c_int32), ('lig_counts', _C.c_void_p),
                ('rec_counts', _C.c_void_p), ('e_ll', _C.c_int64), ('e_rr', _C.c_int64), ('src_ll', _C.c_void_p),
                ('dst_ll', _C.c_void_p), ('src_rr', _C.c_void_p), ('dst_rr', _C.c_void_p), ('he_ll', _C.c_void_p),
                ('he_rr', _C.c_void_p), ('tile_edges', _C.c_int32), ('tile_nodes', _C.c_int32), ('att_block', _C.c_int32),
                ('he_ll_parts', _C.c_void_p), ('he_rr_parts', _C.c_void_p), ('he_ll_cum', _C.c_void_p),
                ('he_rr_cum', _C.c_void_p), ('n_threads', _C.c_int32)]


class _HostCollateIn(_C.Structure):
    _fields_ = [('n_pairs', _C.c_int32)] + [(k, _C.c_void_p) for k in (
        'nl', 'nr', 'el', 'er', 'lig_x', 'lig_new_x', 'lig_res', 'lig_mu', 'rec_x', 'rec_res', 'rec_mu', 'he_ll', 'he_rr',
        'src_ll', 'dst_ll', 'src_rr', 'dst_rr')] + [(k, _C.c_int32) for k in ('tile_edges', 'tile_nodes', 'att_block',
                                                                             'n_threads')]


class _HostCollateOut(_C.Structure):
    _fields_ = [(k, _C.c_void_p) for k in ('lig_x', 'lig_new_x', 'lig_res', 'lig_mu', 'rec_x', 'rec_res', 'rec_mu', 'mu', 'x0',
                                           'res_id', 'src_ll', 'dst_ll', 'src_rr', 'dst_rr')] + [('bad_res', _C.c_int32)]


class _HostPackOut(_C.Structure):
    _fields_ = [(k, _C.c_void_p) for k in ('lig_off', 'rec_off', 'src', 'dst', 'rowptr', 'csc_ptr', 'csc_eid', 'tile_node',
                                           'att_items', 'seg_off', 'edge_perm', 'he', 'he_bf16')] + \
               [(k, _C.c_int32) for k in ('items_cap', 'n_tiles', 'n_att_items', 'max_seg', 'max_degree')]


_host_lib = None


def _native():
    """libequidock_host.so if it has been built (python -m equidock_public_amd.build), else None (numpy path).
    EQD_NATIVE_PACK=0 forces the numpy path (tests compare the two)."""
    global _host_lib
    if os.environ.get('EQD_NATIVE_PACK') == '0':
        return None
    if _host_lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libequidock_host.so')
        if not os.path.exists(path):
            _host_lib = False
        else:
            lib = _C.CDLL(path)
            lib.eqd_host_pack.restype = _C.c_int
            lib.eqd_host_collate_pack.restype = _C.c_int
            _host_lib = lib if lib.eqd_host_pack_abi() == 2 else False
    return _host_lib or None


class _DataView:
    def __init__(self, store):
        self.data = store


class _Indexer:
    def __init__(self, stores):
        self._stores = stores

    def __getitem__(self, key):
        if isinstance(key, tuple):      # ('ligand', 'll', 'ligand') style canonical etype
            key = key[1]
        return _DataView(self._stores[key])


class PairGraph:
    """One (ligand, receptor) pair or a batch of them (same class, like a DGL graph)."""

    NTYPES = ('ligand', 'receptor')
    ETYPES = {'ll': 'ligand', 'rr': 'receptor'}

    def __init__(self, ndata, edata, edges, batch_nodes, batch_edges):
        self._ndata = ndata
        self._edata = edata
        self._edges = edges                 # {'ll': (src, dst), 'rr': (src, dst)} int32, block-local ids
        self._batch_nodes = batch_nodes     # {'ligand': [n...], 'receptor': [n...]}
        self._batch_edges = batch_edges     # {'ll': [e...], 'rr': [e...]}
        self._packed = None

    # ---- DGL-like accessors ------------------------------------------------------------------
    @property
    def nodes(self):
        return _Indexer(self._ndata)

    @property
    def edges(self):
        return _Indexer(self._edata)

    def num_nodes(self, ntype):
        return int(sum(self._batch_nodes[ntype]))

    def num_edges(self, etype):
        return int(sum(self._batch_edges[etype]))

    def batch_num_nodes(self, ntype):
        return torch.tensor(self._batch_nodes[ntype], dtype=torch.int64)

    def batch_num_edges(self, etype):
        return torch.tensor(self._batch_edges[etype], dtype=torch.int64)

    @property
    def batch_size(self):
        return len(self._batch_nodes['ligand'])

    @property
    def device(self):
        return self._ndata['receptor']['x'].device

    def edge_endpoints(self, etype):
        return self._edges[etype]

    def to(self, device):
        device = torch.device(device)
        for store in list(self._ndata.values()) + list(self._edata.values()):
            for k in list(store):
                store[k] = store[k].to(device)
        self._edges = {k: (s.to(device), d.to(device)) for k, (s, d) in self._edges.items()}
        if self._packed is not None:
            self._packed = self._packed.to(device)
        return self

    def pin_memory(self):
        """Called by torch.utils.data.DataLoader(pin_memory=True) on the collated batch (in its pin thread): after it,
        .to('cuda') is a handful of asynchronous copies from page-locked memory."""
        for store in list(self._ndata.values()) + list(self._edata.values()):
            for k in list(store):
                if store[k].device.type == 'cpu':
                    store[k] = store[k].pin_memory()
        self._edges = {k: (s.pin_memory() if s.device.type == 'cpu' else s, d.pin_memory() if d.device.type == 'cpu' else d)
                       for k, (s, d) in self._edges.items()}
        if self._packed is not None:
            self._packed.pin_memory()
        return self

    def with_ligand_coords(self, new_x):
        """A batch that shares everything with this one except the ligand's `new_x` (the reference's fine-tune stage
        writes the first stage's docked ligand back and re-batches, rigid_docking_model.py:667-669, 681-682); the packed
        topology is shared too, only the coordinate block is re-read."""
        nd = {nt: dict(v) for nt, v in self._ndata.items()}
        nd['ligand']['new_x'] = new_x
        g = PairGraph(nd, self._edata, self._edges, self._batch_nodes, self._batch_edges)
        if self._packed is not None:
            import copy
            g._packed = copy.copy(self._packed)
            g._packed.x0 = None
            g._packed._x0_key = None
            g._packed._cstruct = None
        return g

    # ---- packing for the HIP path ---------------------------------------------------------------
    def pack(self):
        """Device-resident kernel layout (topology cached; coordinates re-read every call)."""
        if self._packed is None or _canon(self._packed.device) != _canon(self.device):
            self._packed = PackedGraph.build(self)
        self._packed.refresh_coords(self)
        return self._packed


def _canon(device):
    """(type, index) with the index of a bare 'cuda' resolved, so that torch.device('cuda') == cuda:<current>: a layout
    that was packed on the host and moved with .to('cuda') must not be rebuilt from GPU tensors at the first forward."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        return (device.type, torch.cuda.current_device())
    return (device.type, device.index)


def pair_from_arrays(lig, rec):
    """Build a single-pair graph from two dicts of arrays (numpy or torch) with keys
    x, res_feat, mu_r_norm, src, dst, he and (ligand only) new_x -- the node/edge data the
    reference attaches in hetero_graph_from_sg_l_r_pair (src/utils/train_utils.py:71-82)."""
    def t(a, dtype):
        return torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dtype).contiguous()

    ndata = {
        'ligand': {'res_feat': t(lig['res_feat'], torch.float32).view(-1, 1), 'x': t(lig['x'], torch.float32),
                   'new_x': t(lig['new_x'], torch.float32), 'mu_r_norm': t(lig['mu_r_norm'], torch.float32)},
        'receptor': {'res_feat': t(rec['res_feat'], torch.float32).view(-1, 1), 'x': t(rec['x'], torch.float32),
                     'mu_r_norm': t(rec['mu_r_norm'], torch.float32)},
    }
    edata = {'ll': {'he': t(lig['he'], torch.float32)}, 'rr': {'he': t(rec['he'], torch.float32)}}
    edges = {'ll': (t(lig['src'], torch.int32), t(lig['dst'], torch.int32)),
             'rr': (t(rec['src'], torch.int32), t(rec['dst'], torch.int32))}
    nl, nr = ndata['ligand']['x'].shape[0], ndata['receptor']['x'].shape[0]
    nt_of = PairGraph.ETYPES
    for et, n in (('ll', nl), ('rr', nr)):
        s, d = edges[et]
        if s.numel():
            if int(s.min()) < 0 or int(d.min()) < 0 or int(s.max()) >= n or int(d.max()) >= n:
                raise ValueError(f"edge endpoint out of range for edge type {et}")
        if edata[et]['he'].shape[0] != s.numel():
            raise ValueError(f"'he' has {edata[et]['he'].shape[0]} rows for {s.numel()} {et} edges")
        _check_feature_widths(edata[et]['he'], ndata[nt_of[et]]['mu_r_norm'], et)
    return PairGraph(ndata, edata, edges,
                     {'ligand': [nl], 'receptor': [nr]},
                     {'ll': [int(edges['ll'][0].numel())], 'rr': [int(edges['rr'][0].numel())]})


HE_WIDTH = 27        # 15 distance RBFs + 12 orientation features (src/utils/protein_utils.py:71-86, 380-390)
MU_WIDTH = 5         # surface-aware features, sigma in {1, 2, 5, 10, 30} (src/utils/protein_utils.py:351-359)


def _check_feature_widths(he, mu, et):
    """The kernels (and the native host pack) address `he` rows as 27 floats and `mu_r_norm` rows as 5."""
    if he.dim() != 2 or he.shape[1] != HE_WIDTH:
        raise ValueError(f"'he' of edge type {et} must be [E, {HE_WIDTH}] (input_edge_feats_dim of the HIP path), got "
                         f"{tuple(he.shape)}")
    if mu.dim() != 2 or mu.shape[1] != MU_WIDTH:
        raise ValueError(f"'mu_r_norm' must be [n, {MU_WIDTH}], got {tuple(mu.shape)}")


def batch(graphs):
    """dgl.batch equivalent (src/utils/train_utils.py:98): concatenate node/edge data per type,
    offset edge endpoints by the cumulative node counts, remember per-pair counts."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("empty batch")
    ndata = {nt: {k: torch.cat([g._ndata[nt][k] for g in graphs], 0) for k in graphs[0]._ndata[nt]}
             for nt in PairGraph.NTYPES}
    edata = {et: {k: torch.cat([g._edata[et][k] for g in graphs], 0) for k in graphs[0]._edata[et]}
             for et in PairGraph.ETYPES}
    edges = {}
    for et, nt in PairGraph.ETYPES.items():
        off, ss, dd = 0, [], []
        for g in graphs:
            s, d = g._edges[et]
            ss.append(s + off)
            dd.append(d + off)
            off += g.num_nodes(nt)
        edges[et] = (torch.cat(ss).to(torch.int32), torch.cat(dd).to(torch.int32))
    bn = {nt: [n for g in graphs for n in g._batch_nodes[nt]] for nt in PairGraph.NTYPES}
    be = {et: [n for g in graphs for n in g._batch_edges[et]] for et in PairGraph.ETYPES}
    return PairGraph(ndata, edata, edges, bn, be)


def unbatch(g):
    """dgl.unbatch equivalent: per-pair graphs carrying (views of) all current node/edge data."""
    outs = []
    noff = {nt: 0 for nt in PairGraph.NTYPES}
    eoff = {et: 0 for et in PairGraph.ETYPES}
    for i in range(g.batch_size):
        nn = {nt: g._batch_nodes[nt][i] for nt in PairGraph.NTYPES}
        ne = {et: g._batch_edges[et][i] for et in PairGraph.ETYPES}
        ndata = {nt: {k: v[noff[nt]:noff[nt] + nn[nt]] for k, v in g._ndata[nt].items()} for nt in PairGraph.NTYPES}
        edata = {et: {k: v[eoff[et]:eoff[et] + ne[et]] for k, v in g._edata[et].items()} for et in PairGraph.ETYPES}
        edges = {}
        for et, nt in PairGraph.ETYPES.items():
            s, d = g._edges[et]
            edges[et] = (s[eoff[et]:eoff[et] + ne[et]] - noff[nt], d[eoff[et]:eoff[et] + ne[et]] - noff[nt])
        outs.append(PairGraph(ndata, edata, edges, {nt: [nn[nt]] for nt in nn}, {et: [ne[et]] for et in ne}))
        for nt in noff:
            noff[nt] += nn[nt]
        for et in eoff:
            eoff[et] += ne[et]
    return outs


def uniform_rotation_translation(translation_interval):
    """src/utils/protein_utils.py:15-23 (UniformRotation_Translation): a uniformly random rotation (scipy) and a translation
    of uniformly random length in [0, translation_interval) along a random direction, drawn from numpy's GLOBAL generator
    in the reference's order - the same np.random.seed gives the reference's draws.  Returns float32 (3, 3), (1, 3)."""
    from scipy.spatial.transform import Rotation
    rotation_matrix = Rotation.random(num=1).as_matrix().squeeze()
    t = np.random.randn(1, 3)
    t = t / np.sqrt(np.sum(t * t))
    length = np.random.uniform(low=0, high=translation_interval)
    t = t * length
    return rotation_matrix.astype(np.float32), t.astype(np.float32)


def augment_ligand(batch, rotations, translations, pocket_coors_ligand_list=None):
    """The reference's per-item random rigid perturbation of the ligand (src/utils/db5_data.py:195-204), for a whole
    batch in one launch ON THE DEVICE (eqd_rigid_augment): ligand `new_x` <- R_p (x - mean_p) + t_p, written in place, and
    - when given - the pairs' ligand pocket coordinates under the same maps (returned as a list).  rotations [B, 3, 3],
    translations [B, 3] (or [B, 1, 3]): host or device, e.g. B draws of uniform_rotation_translation()."""
    import ctypes as C
    from . import _lib
    lib = _lib.load_library()
    packed = batch.pack()
    dev = packed.x0.device
    B = packed.n_pairs
    R = torch.as_tensor(np.asarray(rotations) if not torch.is_tensor(rotations) else rotations).to(dev, torch.float32) \
        .reshape(B, 9).contiguous()
    t = torch.as_tensor(np.asarray(translations) if not torch.is_tensor(translations) else translations) \
        .to(dev, torch.float32).reshape(B, 3).contiguous()
    x = _lib.require_device(batch._ndata['ligand']['x'].to(torch.float32).contiguous(), "ligand 'x'")
    new_x = batch._ndata['ligand']['new_x']
    if new_x.dtype != torch.float32 or not new_x.is_contiguous() or new_x.device != x.device:
        new_x = torch.empty_like(x)
        batch._ndata['ligand']['new_x'] = new_x
    off = pin = pout = None
    if pocket_coors_ligand_list is not None:
        counts = [int(p.shape[0]) for p in pocket_coors_ligand_list]
        if len(counts) != B:
            raise ValueError("one pocket array per pair expected")
        pin = torch.cat([torch.as_tensor(p).reshape(-1, 3) for p in pocket_coors_ligand_list]).to(dev, torch.float32) \
            .contiguous()
        pout = torch.empty_like(pin)
        off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32).to(dev)
    gs = packed.c_struct()
    with _lib.device_guard(dev):
        _lib.check(lib.eqd_rigid_augment(C.byref(gs), _lib.ptr(x), _lib.ptr(R), _lib.ptr(t), _lib.ptr(new_x), _lib.ptr(off),
                                         _lib.ptr(pin), _lib.ptr(pout), _lib.stream_ptr(dev)))
    new_x.add_(0)       # bump the version counter: the packed layout re-reads [new_x ; x] at the next forward
    if pout is None:
        return None
    return list(torch.split(pout, counts))


def from_dgl(g):
    """Adapter from the reference's input object: a (batched) DGL heterograph as built by
    hetero_graph_from_sg_l_r_pair + dgl.batch (src/utils/train_utils.py:61-100) -> PairGraph.

    Duck-typed - DGL itself is not imported (it is a third-party pin that is absent here): only the accessors the
    reference's own model file uses are needed, `g.nodes[ntype].data[...]`, `g.edges[canonical_etype].data['he']`,
    `g.batch_num_nodes(ntype)`, `g.batch_num_edges(etype)` and the endpoint query `g.edges(etype=...)` (DGL's
    EdgeView is callable; `g.all_edges(etype=...)` is tried as well).  The empty 'cross' edge types of the reference's
    heterograph are ignored.  Tensors are shared, not copied; the result packs bit-identically to batch_pairs() on the
    same pairs (tests/test_abi_and_graph.py, through the oracle's DGL stand-in)."""
    if isinstance(g, PairGraph):
        return g
    cet = {'ll': ('ligand', 'll', 'ligand'), 'rr': ('receptor', 'rr', 'receptor')}
    try:
        nd = {'ligand': {k: g.nodes['ligand'].data[k] for k in ('res_feat', 'x', 'new_x', 'mu_r_norm')},
              'receptor': {k: g.nodes['receptor'].data[k] for k in ('res_feat', 'x', 'mu_r_norm')}}
        ed = {et: {'he': g.edges[c].data['he']} for et, c in cet.items()}
    except (KeyError, AttributeError, TypeError) as e:
        raise TypeError("expected a PairGraph or a DGL heterograph with node types ligand/receptor carrying res_feat, x, "
                        f"mu_r_norm (+ new_x on the ligand) and edge types ll/rr carrying he: {e!r}")

    def endpoints(c):
        for call in (lambda: g.edges(etype=c), lambda: g.all_edges(etype=c), lambda: g.edges(etype=c[1])):
            try:
                s, d = call()[:2]
                return s, d
            except (TypeError, AttributeError, KeyError):
                continue
        raise TypeError(f"cannot query the endpoints of edge type {c}")
    edges = {}
    for et, c in cet.items():
        s, d = endpoints(c)
        edges[et] = (torch.as_tensor(s).to(torch.int32).contiguous(), torch.as_tensor(d).to(torch.int32).contiguous())
        if ed[et]['he'].shape[0] != edges[et][0].numel():
            raise ValueError(f"'he' has {ed[et]['he'].shape[0]} rows for {edges[et][0].numel()} {et} edges")
    bn = {nt: [int(v) for v in g.batch_num_nodes(nt)] for nt in PairGraph.NTYPES}
    be = {et: [int(v) for v in g.batch_num_edges(c)] for et, c in cet.items()}
    for nt in PairGraph.NTYPES:
        nd[nt] = {k: (v.to(torch.float32) if v.dtype != torch.float32 else v) for k, v in nd[nt].items()}
        nd[nt]['res_feat'] = nd[nt]['res_feat'].reshape(-1, 1)
        if sum(bn[nt]) != nd[nt]['x'].shape[0]:
            raise ValueError(f"batch_num_nodes('{nt}') does not add up to the node data rows")
    return PairGraph(nd, ed, edges, bn, be)


def batch_pairs(pairs, n_threads=4):
    """List of (ligand_dict, receptor_dict) of arrays -> batched PairGraph: the collate function of the drop-in (the
    reference builds one dgl.heterograph per pair and calls dgl.batch, src/utils/train_utils.py:61-100).

    With libequidock_host.so built this is ONE native call (eqd_host_collate_pack) that goes from the per-pair arrays
    straight to the batch's node arrays AND the packed kernel layout (so `.pack()` afterwards is free); otherwise - or
    with EQD_NATIVE_PACK=0 - the per-pair construction + batch() below.  Both give bit-identical batches
    (tests/test_abi_and_graph.py)."""
    pairs = list(pairs)
    lib = _native()
    if lib is not None and pairs:
        return _batch_pairs_native(lib, pairs, n_threads)
    return batch([pair_from_arrays(l, r) for l, r in pairs])


def _np(a, dtype):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _batch_pairs_native(lib, pairs, n_threads):
    B = len(pairs)
    keep = []           # per-pair arrays must stay alive until the call returns
    cols = {k: [] for k in ('lig_x', 'lig_new_x', 'lig_res', 'lig_mu', 'rec_x', 'rec_res', 'rec_mu', 'he_ll', 'he_rr',
                            'src_ll', 'dst_ll', 'src_rr', 'dst_rr')}
    nl, nr, el, er = [], [], [], []
    for lig, rec in pairs:
        a = dict(lig_x=_np(lig['x'], np.float32), lig_new_x=_np(lig['new_x'], np.float32),
                 lig_res=_np(lig['res_feat'], np.float32).reshape(-1), lig_mu=_np(lig['mu_r_norm'], np.float32),
                 rec_x=_np(rec['x'], np.float32), rec_res=_np(rec['res_feat'], np.float32).reshape(-1),
                 rec_mu=_np(rec['mu_r_norm'], np.float32), he_ll=_np(lig['he'], np.float32), he_rr=_np(rec['he'], np.float32),
                 src_ll=_np(lig['src'], np.int32), dst_ll=_np(lig['dst'], np.int32),
                 src_rr=_np(rec['src'], np.int32), dst_rr=_np(rec['dst'], np.int32))
        n_l, n_r, e_l, e_r = a['lig_x'].shape[0], a['rec_x'].shape[0], a['src_ll'].shape[0], a['src_rr'].shape[0]
        ok = (a['lig_x'].shape == (n_l, 3) and a['lig_new_x'].shape == (n_l, 3) and a['lig_res'].shape == (n_l,)
              and a['lig_mu'].shape == (n_l, MU_WIDTH) and a['rec_x'].shape == (n_r, 3) and a['rec_res'].shape == (n_r,)
              and a['rec_mu'].shape == (n_r, MU_WIDTH) and a['he_ll'].shape == (e_l, HE_WIDTH)
              and a['he_rr'].shape == (e_r, HE_WIDTH) and a['dst_ll'].shape == (e_l,) and a['dst_rr'].shape == (e_r,))
        if not ok:
            raise ValueError("inconsistent array shapes in a (ligand, receptor) pair: x / new_x [n, 3], res_feat [n(, 1)], "
                             f"mu_r_norm [n, {MU_WIDTH}], src / dst [e], he [e, {HE_WIDTH}]")
        keep.append(a)
        for k in cols:
            cols[k].append(a[k].ctypes.data)
        nl.append(n_l); nr.append(n_r); el.append(e_l); er.append(e_r)
    NL, NR, EL, ER = sum(nl), sum(nr), sum(el), sum(er)
    n, E = NL + NR, EL + ER
    p = PackedGraph()
    p.n_pairs, p.n_lig, p.n_rec, p.n_nodes = B, NL, NR, n
    items_cap = XCD_CLASSES * sum((c + ATT_BLOCK - 1) // ATT_BLOCK for c in nl + nr)      # see _xcd_interleave
    flats, views = PackedGraph._alloc_native_buffers(B, n, E, items_cap)
    f32, i32 = torch.float32, torch.int32
    nd = {'ligand': {'res_feat': torch.empty(NL, 1, dtype=f32), 'x': torch.empty(NL, 3, dtype=f32),
                     'new_x': torch.empty(NL, 3, dtype=f32), 'mu_r_norm': torch.empty(NL, 5, dtype=f32)},
          'receptor': {'res_feat': torch.empty(NR, 1, dtype=f32), 'x': torch.empty(NR, 3, dtype=f32),
                       'mu_r_norm': torch.empty(NR, 5, dtype=f32)}}
    x0 = torch.empty(n, 3, dtype=f32)
    edges = {'ll': (torch.empty(EL, dtype=i32), torch.empty(EL, dtype=i32)),
             'rr': (torch.empty(ER, dtype=i32), torch.empty(ER, dtype=i32))}
    counts = {k: np.asarray(v, dtype=np.int64) for k, v in (('nl', nl), ('nr', nr), ('el', el), ('er', er))}
    tabs = {k: (_C.c_void_p * B)(*v) for k, v in cols.items()}
    ci = _HostCollateIn()
    ci.n_pairs = B
    for k, v in counts.items():
        setattr(ci, k, v.ctypes.data)
    for k, v in tabs.items():
        setattr(ci, k, _C.cast(v, _C.c_void_p))
    ci.tile_edges, ci.tile_nodes, ci.att_block, ci.n_threads = TILE_EDGES, TILE_NODES, ATT_BLOCK, int(n_threads)
    co = _HostCollateOut()
    co.lig_x, co.lig_new_x = nd['ligand']['x'].data_ptr(), nd['ligand']['new_x'].data_ptr()
    co.lig_res, co.lig_mu = nd['ligand']['res_feat'].data_ptr(), nd['ligand']['mu_r_norm'].data_ptr()
    co.rec_x, co.rec_res, co.rec_mu = (nd['receptor'][k].data_ptr() for k in ('x', 'res_feat', 'mu_r_norm'))
    co.mu, co.x0, co.res_id = views['mu_r_norm'].data_ptr(), x0.data_ptr(), views['res_id'].data_ptr()
    co.src_ll, co.dst_ll = edges['ll'][0].data_ptr(), edges['ll'][1].data_ptr()
    co.src_rr, co.dst_rr = edges['rr'][0].data_ptr(), edges['rr'][1].data_ptr()
    hout = PackedGraph._host_out(views, items_cap)
    rc_ = lib.eqd_host_collate_pack(_C.byref(ci), _C.byref(co), _C.byref(hout))
    PackedGraph._check_native_rc(rc_, hout)
    if co.bad_res == 1:
        raise ValueError("res_feat must hold residue ids 0..20 (nn.Embedding(21, .), rigid_docking_model.py:382)")
    if co.bad_res == 2:
        raise ValueError("mu_r_norm must be > 0 (the model takes its log, rigid_docking_model.py:469)")
    p._finish_native(flats, views, hout, E, nl, nr)
    # edge features of the container: views into the packed array when the edges were already destination-sorted (they
    # are as the reference builds them, src/utils/protein_utils.py:339-346); re-ordered copies otherwise
    perm = p.edge_perm
    if E == 0 or bool((perm == torch.arange(E, dtype=perm.dtype)).all()):
        he_ll, he_rr = p.he[:EL], p.he[EL:]
    else:
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(E, dtype=perm.dtype)
        he_all = p.he[inv]
        he_ll, he_rr = he_all[:EL], he_all[EL:]
    g = PairGraph(nd, {'ll': {'he': he_ll}, 'rr': {'he': he_rr}}, edges, {'ligand': nl, 'receptor': nr},
                  {'ll': el, 'rr': er})
    p.x0 = x0
    lx, rx = nd['ligand']['new_x'], nd['receptor']['x']
    p._x0_key = (lx.data_ptr(), lx._version, rx.data_ptr(), rx._version, lx.device)
    g._packed = p
    return g


XCD_CLASSES = 8      # MI355X: 8 XCDs with a private 4 MiB L2 each; block b of a launch is placed on XCD b % 8


def _xcd_interleave(items):
    """Order the attention work list so that the blocks which stream the SAME partner rows run on the same XCD.

    Every item (query block, partner range) re-reads its partner's key / value rows; the rows of one (pair, direction) are
    shared by all of that direction's items.  Workgroup b lands on XCD b % 8 (observed dispatch rule, a speed matter only),
    so the list is laid out as 8 interleaved queues, slot 8 k + c = k-th item of queue c: the partner rows are then fetched
    from HBM into ONE L2 instead of up to eight (measured 5.7-6.3 x the algorithmic bytes before, profiles/r02_z_traffic).
    The GROUPS (all blocks of one (pair, direction): same partner range) are sorted by decreasing partner size and dealt
    into 8 lists in snake order (0..7, 7..0, ...), so that every list holds big and small partners alike; the lists are
    concatenated and cut into 8 queues of equal COST (cost ~ block x partner rows; a group spills into the next queue
    where it must - one pair still spreads over all XCDs), each queue then runs its biggest partners first, and the
    queues are padded with empty items (0, 0, 0, 0) to equal length.  (Round 2 cut the size-sorted list directly: with
    ragged sizes one XCD then got only the biggest partners - few long work items, the last round of its 64 workgroup
    slots mostly idle - and another only the smallest; 77 % -> 93 % of the ideal makespan for the DB5.5 size
    distribution in a slot model, profiles/r03_attention_schedule.txt.  Uniform batches are unchanged.)  The native
    packer (csrc_host/eqd_host_pack.cpp) does the same in integers, bit for bit."""
    # groups = runs of CONSECUTIVE items with the same partner range, exactly as the native packer forms them (two
    # degenerate segments with equal ranges that are not neighbours stay two groups in both)
    groups, order = {}, []
    for it in items:
        if not order or (order[-1][0], order[-1][1]) != (it[2], it[3]):
            order.append((it[2], it[3], len(order)))
            groups[order[-1]] = []
        groups[order[-1]].append(it)
    order = sorted(order, key=lambda k: -(k[1] - k[0]))      # stable
    lists = [[] for _ in range(XCD_CLASSES)]
    for i, key in enumerate(order):
        r = i % (2 * XCD_CLASSES)
        lists[r if r < XCD_CLASSES else 2 * XCD_CLASSES - 1 - r].append(key)
    items = [it for lst in lists for key in lst for it in groups[key]]
    total = sum(it[3] - it[2] for it in items)
    queues = [[] for _ in range(XCD_CLASSES)]
    c, acc = 0, 0
    for it in items:
        queues[c].append(it)
        acc += it[3] - it[2]
        if c < XCD_CLASSES - 1 and acc * XCD_CLASSES >= total * (c + 1):
            c += 1
    queues = [sorted(q, key=lambda it: -(it[3] - it[2])) for q in queues]      # stable: biggest partner first
    depth = max(len(q) for q in queues) if items else 0
    out = np.zeros((depth * XCD_CLASSES, 4), dtype=np.int32)
    for c, q in enumerate(queues):
        if q:
            out[c:len(q) * XCD_CLASSES:XCD_CLASSES] = np.asarray(q, dtype=np.int32)
    return out


class PackedGraph:
    """Kernel-side layout of a batch; every tensor lives on the batch's device.

    int32: lig_off[B+1], rec_off[B+1] (block-local), src[E], dst[E] (global node ids, dst-sorted),
           rowptr[N+1], csc_ptr[N+1], csc_eid[E], tile_node[T+1], att_items[I,4], res_id[N],
           seg_off[2B+1] (global node offsets of the 2B segments: B ligands then B receptors)
    fp32 : mu_r_norm[N,5], he[E,27], x0[N,3] (ligand new_x rows then receptor x rows)
    bf16 : he_bf16[E,32] (he rounded to nearest even, 5 zero columns; stored as int16 bit patterns)
    """

    INT_FIELDS = ('lig_off', 'rec_off', 'src', 'dst', 'rowptr', 'csc_ptr', 'csc_eid', 'tile_node',
                  'att_items', 'res_id', 'seg_off')

    def __init__(self):
        self.device = None
        self._cstruct = None
        self.ws_sizes = {}

    def c_struct(self):
        """EqdGraph view of this batch (cached; only the coordinate pointer changes between calls)."""
        from . import _lib
        if self._cstruct is None:
            self._cstruct = _lib.graph_struct(self)
        else:
            _lib.require_device(self.x0, 'graph.x0')
            self._cstruct.x0 = self.x0.data_ptr()
        return self._cstruct

    @staticmethod
    def build(g):
        p = PackedGraph()
        dev = g.device
        nl, nr = g.num_nodes('ligand'), g.num_nodes('receptor')
        n = nl + nr
        B = g.batch_size
        p.n_pairs, p.n_lig, p.n_rec, p.n_nodes = B, nl, nr, n
        for et, nt in PairGraph.ETYPES.items():
            _check_feature_widths(g._edata[et]['he'], g._ndata[nt]['mu_r_norm'], et)
        native = _native() if (dev.type == 'cpu' and n > 0) else None
        if native is not None:
            return PackedGraph._build_native(native, g, p)
        lig_off = np.concatenate([[0], np.cumsum(g._batch_nodes['ligand'])]).astype(np.int32)
        rec_off = np.concatenate([[0], np.cumsum(g._batch_nodes['receptor'])]).astype(np.int32)

        srcs, dsts, perms, eoff = [], [], [], 0
        for et, base in (('ll', 0), ('rr', nl)):
            s, d = g._edges[et]
            s = s.detach().cpu().numpy().astype(np.int64)
            d = d.detach().cpu().numpy().astype(np.int64)
            perm = np.arange(len(d), dtype=np.int64)
            if len(d) > 1 and np.any(d[1:] < d[:-1]):
                perm = np.argsort(d, kind='stable')
                s, d = s[perm], d[perm]
            srcs.append(s + base)
            dsts.append(d + base)
            perms.append(perm + eoff)
            eoff += len(d)
        src = np.concatenate(srcs).astype(np.int32)
        dst = np.concatenate(dsts).astype(np.int32)
        perm = np.concatenate(perms)
        E = len(src)
        p.n_edges = E
        deg = np.bincount(dst, minlength=n).astype(np.int64)
        if E and deg.max() > TILE_EDGES:
            raise ValueError(f"in-degree {int(deg.max())} exceeds the supported maximum of {TILE_EDGES} "
                             "(the reference caps it at graph_max_neighbor=10, src/utils/args.py:47)")
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        order = np.argsort(src, kind='stable').astype(np.int32)
        csc_ptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]).astype(np.int32)

        # node-aligned edge tiles
        # greedy: a tile takes nodes while it stays within TILE_EDGES edges and TILE_NODES nodes.  The end of the tile
        # that starts at node i is computed for every i at once (searchsorted on the edge prefix sums); the tiling is
        # then the chain 0 -> nxt[0] -> nxt[nxt[0]] ... (one list lookup per tile instead of Python work per node)
        if n:
            cs = rowptr.astype(np.int64)
            nxt = np.searchsorted(cs, cs[:-1] + TILE_EDGES, side='right') - 1
            idx = np.arange(n, dtype=np.int64)
            nxt = np.minimum(np.minimum(nxt, idx + TILE_NODES), n)
            nxt = np.maximum(nxt, idx + 1).tolist()
            tiles, i = [0], 0
            while i < n:
                i = nxt[i]
                tiles.append(i)
        else:
            tiles = [0, 0]
        tile_node = np.asarray(tiles, dtype=np.int32)
        p.n_tiles = len(tiles) - 1

        # segments: B ligand segments then B receptor segments (global node ids)
        seg_off = np.concatenate([lig_off[:-1], nl + rec_off]).astype(np.int32)
        items = []
        for b in range(B):
            l0, l1 = int(lig_off[b]), int(lig_off[b + 1])
            r0, r1 = nl + int(rec_off[b]), nl + int(rec_off[b + 1])
            for (a0, a1, o0, o1) in ((l0, l1, r0, r1), (r0, r1, l0, l1)):
                for s0 in range(a0, a1, ATT_BLOCK):
                    items.append((s0, min(s0 + ATT_BLOCK, a1), o0, o1))
        att_items = _xcd_interleave(items)
        p.n_att_items = att_items.shape[0]
        p.max_seg = int(max(max(g._batch_nodes['ligand']), max(g._batch_nodes['receptor'])))

        res = torch.cat([g._ndata['ligand']['res_feat'].view(-1), g._ndata['receptor']['res_feat'].view(-1)])
        res_id = res.detach().cpu().numpy().astype(np.int32)
        if n and (res_id.min() < 0 or res_id.max() > 20):
            raise ValueError("res_feat must hold residue ids 0..20 (nn.Embedding(21, .), rigid_docking_model.py:382)")

        def ti(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        p.lig_off, p.rec_off, p.src, p.dst = ti(lig_off), ti(rec_off), ti(src), ti(dst)
        p.rowptr, p.csc_ptr, p.csc_eid = ti(rowptr), ti(csc_ptr), ti(order)
        p.tile_node, p.att_items, p.res_id, p.seg_off = ti(tile_node), ti(att_items), ti(res_id), ti(seg_off)
        p.edge_perm = torch.from_numpy(perm).to(dev)
        he = torch.cat([g._edata['ll']['he'], g._edata['rr']['he']], 0).to(torch.float32)
        p.he = he[p.edge_perm].contiguous() if E else he.contiguous()
        # bf16 copy for the storage_bf16 mode: 32 columns per edge (27 used), i.e. 64-byte rows
        hb = torch.zeros(max(E, 1), 32, dtype=torch.bfloat16, device=p.he.device)
        if E:
            hb[:, :27] = p.he.to(torch.bfloat16)
        p.he_bf16 = hb.view(torch.int16).contiguous()
        p.mu_r_norm = torch.cat([g._ndata['ligand']['mu_r_norm'], g._ndata['receptor']['mu_r_norm']], 0) \
            .to(torch.float32).contiguous()
        if n and not float(p.mu_r_norm.min()) > 0.0:      # `not >` also rejects NaN
            raise ValueError("mu_r_norm must be > 0 (the model takes its log, rigid_docking_model.py:469)")
        p.x0 = None
        p.device = dev
        p.lig_counts = list(g._batch_nodes['ligand'])
        p.rec_counts = list(g._batch_nodes['receptor'])
        p._flats = None
        return p.consolidate()

    @staticmethod
    def _build_native(lib, g, p):
        """Same layout through libequidock_host.so: one pass of C++ loops straight into the per-dtype buffers that
        consolidate() would otherwise assemble from ~25 numpy arrays (bit-identical results)."""
        B, nl, nr = p.n_pairs, p.n_lig, p.n_rec
        n = nl + nr

        def c32(t):
            return t.detach().to(torch.int32).contiguous()
        s_ll, d_ll = (c32(t) for t in g._edges['ll'])
        s_rr, d_rr = (c32(t) for t in g._edges['rr'])
        he_ll = g._edata['ll']['he'].detach().to(torch.float32).contiguous()
        he_rr = g._edata['rr']['he'].detach().to(torch.float32).contiguous()
        E = s_ll.numel() + s_rr.numel()
        lc = torch.tensor(list(g._batch_nodes['ligand']), dtype=torch.int64)
        rc = torch.tensor(list(g._batch_nodes['receptor']), dtype=torch.int64)
        res = torch.cat([g._ndata['ligand']['res_feat'].view(-1), g._ndata['receptor']['res_feat'].view(-1)]) \
            .detach().to(torch.int32)
        if int(res.min()) < 0 or int(res.max()) > 20:
            raise ValueError("res_feat must hold residue ids 0..20 (nn.Embedding(21, .), rigid_docking_model.py:382)")
        mu = torch.cat([g._ndata['ligand']['mu_r_norm'], g._ndata['receptor']['mu_r_norm']], 0).to(torch.float32)
        if not float(mu.min()) > 0.0:      # `not >` also rejects NaN
            raise ValueError("mu_r_norm must be > 0 (the model takes its log, rigid_docking_model.py:469)")
        items_cap = XCD_CLASSES * sum((int(c) + ATT_BLOCK - 1) // ATT_BLOCK for c in list(lc) + list(rc))
        flats, views = PackedGraph._alloc_native_buffers(B, n, E, items_cap)
        views['res_id'].copy_(res)
        views['mu_r_norm'].copy_(mu)
        hin = _HostPackIn(B, nl, nr, lc.data_ptr(), rc.data_ptr(), s_ll.numel(), s_rr.numel(), s_ll.data_ptr(),
                          d_ll.data_ptr(), s_rr.data_ptr(), d_rr.data_ptr(), he_ll.data_ptr(), he_rr.data_ptr(), TILE_EDGES,
                          TILE_NODES, ATT_BLOCK, None, None, None, None, 4)
        hout = PackedGraph._host_out(views, items_cap)
        rc_ = lib.eqd_host_pack(_C.byref(hin), _C.byref(hout))
        PackedGraph._check_native_rc(rc_, hout)
        p._finish_native(flats, views, hout, E, list(g._batch_nodes['ligand']), list(g._batch_nodes['receptor']))
        return p

    @staticmethod
    def _alloc_native_buffers(B, n, E, items_cap):
        """per-dtype buffers with 64-byte aligned slices, laid out here and filled by the host library"""
        spec = {
            torch.int32: [('lig_off', (B + 1,)), ('rec_off', (B + 1,)), ('src', (E,)), ('dst', (E,)), ('rowptr', (n + 1,)),
                          ('csc_ptr', (n + 1,)), ('csc_eid', (E,)), ('tile_node', (n + 2,)), ('att_items', (items_cap, 4)),
                          ('res_id', (n,)), ('seg_off', (2 * B + 1,))],
            torch.int64: [('edge_perm', (E,))],
            torch.float32: [('he', (E, 27)), ('mu_r_norm', (n, 5))],
            torch.int16: [('he_bf16', (max(E, 1), 32))],
        }
        flats, views = {}, {}
        for dt, fields in spec.items():
            esz = torch.empty(0, dtype=dt).element_size()
            align = 64 // esz
            offs, total = [], 0
            for _, shape in fields:
                offs.append(total)
                total += (int(np.prod(shape)) + align - 1) // align * align
            flat = torch.empty(max(total, 1), dtype=dt)
            layout = []
            for (k, shape), o in zip(fields, offs):
                views[k] = flat[o:o + int(np.prod(shape))].view(shape)
                layout.append((k, o, tuple(shape)))
            flats[dt] = (flat, layout)
        return flats, views

    @staticmethod
    def _host_out(views, items_cap):
        hout = _HostPackOut()
        for k in ('lig_off', 'rec_off', 'src', 'dst', 'rowptr', 'csc_ptr', 'csc_eid', 'tile_node', 'att_items', 'seg_off',
                  'edge_perm', 'he', 'he_bf16'):
            setattr(hout, k, views[k].data_ptr())
        hout.items_cap = items_cap
        return hout

    @staticmethod
    def _check_native_rc(rc_, hout):
        if rc_ == 1:
            raise ValueError(f"in-degree {hout.max_degree} exceeds the supported maximum of {TILE_EDGES} "
                             "(the reference caps it at graph_max_neighbor=10, src/utils/args.py:47)")
        if rc_ != 0:
            raise ValueError(f"native pack failed ({rc_}): edge endpoint out of range or inconsistent pair sizes")

    def _finish_native(self, flats, views, hout, E, lig_counts, rec_counts):
        p = self
        p.n_edges, p.n_tiles, p.n_att_items, p.max_seg = E, hout.n_tiles, hout.n_att_items, hout.max_seg
        # the two lists with data-dependent lengths are shorter than their upper bounds: re-slice the views
        for k, shape in (('tile_node', (p.n_tiles + 1,)), ('att_items', (p.n_att_items, 4))):
            views[k] = views[k].reshape(-1)[:int(np.prod(shape))].view(shape)
        for dt, (flat, layout) in flats.items():
            flats[dt] = (flat, [(k, o, tuple(views[k].shape)) for k, o, _ in layout])
        for k, v in views.items():
            setattr(p, k, v)
        p.x0 = None
        p.device = torch.device('cpu')
        p.lig_counts = list(lig_counts)
        p.rec_counts = list(rec_counts)
        p._flats = flats

    def refresh_coords(self, g):
        """x0 = [ligand new_x ; receptor x], re-read at every forward (the training loop re-draws new_x per batch); the
        concatenation is skipped while both tensors are the ones of the last call and were not written to in between."""
        lx, rx = g._ndata['ligand']['new_x'], g._ndata['receptor']['x']
        key = (lx.data_ptr(), lx._version, rx.data_ptr(), rx._version, lx.device)
        if self.x0 is not None and getattr(self, '_x0_key', None) == key:
            return
        self.x0 = torch.cat([lx, rx], 0).to(torch.float32).contiguous()
        self._x0_key = key

    def _tensor_items(self):
        return [(k, v) for k, v in self.__dict__.items() if torch.is_tensor(v) and k != 'x0']

    def consolidate(self):
        """Re-home every tensor of the layout as a 64-byte aligned slice of ONE buffer per dtype (done once, where the
        layout is built - e.g. in a DataLoader worker), so that pinning and the H->D move are one copy per dtype with no
        re-staging on the consumer's side."""
        items = self._tensor_items()
        if not items or any(v.device != items[0][1].device for _, v in items):
            return self
        by_dtype = {}
        for k, v in items:
            by_dtype.setdefault(v.dtype, []).append((k, v))
        flats = {}
        for dt, group in by_dtype.items():
            esz = group[0][1].element_size()
            align = max(1, 64 // esz)
            offs, total = [], 0
            for _, v in group:
                offs.append(total)
                total += (v.numel() + align - 1) // align * align
            flat = torch.empty(max(total, 1), dtype=dt, device=group[0][1].device)
            layout = []
            for (k, v), o in zip(group, offs):
                flat[o:o + v.numel()] = v.reshape(-1)
                setattr(self, k, flat[o:o + v.numel()].view(v.shape))
                layout.append((k, o, tuple(v.shape)))
            flats[dt] = (flat, layout)
        self._flats = flats
        return self

    def _rebind(self, moved):
        for dt, (flat, layout) in moved.items():
            for k, o, shape in layout:
                n = 1
                for d in shape:
                    n *= d
                setattr(self, k, flat[o:o + n].view(shape))
        self._flats = moved
        self._cstruct = None

    def pin_memory(self):
        """torch.utils.data.DataLoader(pin_memory=True) calls this on custom batch objects (in its pin thread)."""
        if getattr(self, '_flats', None) is None:
            self.consolidate()
        if getattr(self, '_flats', None) is not None and self.he.device.type == 'cpu':
            self._rebind({dt: (flat.pin_memory(), layout) for dt, (flat, layout) in self._flats.items()})
        return self

    def to(self, device):
        """Move the layout to `device`: one copy per dtype (see consolidate; pinned staging when the source is pageable
        host memory and the target a GPU), so a batch that was packed in a DataLoader worker (collate_fn:
        batch_pairs(...).pack()) costs three H->D copies instead of one per field."""
        device = torch.device(device)
        if getattr(self, '_flats', None) is None:
            self.consolidate()
        if getattr(self, '_flats', None) is not None:
            moved = {}
            for dt, (flat, layout) in self._flats.items():
                if device.type == 'cuda' and flat.device.type == 'cpu' and not flat.is_pinned():
                    flat = flat.pin_memory()
                moved[dt] = (flat.to(device, non_blocking=True), layout)
            self._rebind(moved)
        else:
            for k, v in self._tensor_items():
                setattr(self, k, v.to(device))
        if self.x0 is not None:
            self.x0 = self.x0.to(device)
        self.device = device
        self._cstruct = None
        return self
