"""ORACLE-ONLY TOOL -- never shipped to the GPU box's product path, never imported by
`equidock_public_amd`.

Minimal stand-in for the `dgl` package (DGL 0.7.0 is pinned by the reference,
requirements.txt:5, and is absent from this image).  It restates exactly the DGL
semantics that /root/reference/src/model/rigid_docking_model.py relies on
 so that the reference model file can be imported
and executed UNMODIFIED on CPU to generate golden vectors (oracle/make_golden.py):

  * g.nodes[ntype].data / g.edges[etype or (s, etype, d)].data   dict views
  * g.local_scope()          writes inside the scope do not persist
  * g.apply_edges(fn.u_sub_v(a, b, out), etype)   edata[out] = ndata[a][src] - ndata[b][dst]
  * g.apply_edges(udf, etype) with edges.src[k] / edges.dst[k] gathers
  * g.update_all(fn.copy_edge(f, 'm'), fn.mean('m', out), etype)
        per-destination mean over in-edges, zeros for in-degree 0
  * g.batch_num_nodes(ntype), g.batch_num_edges(etype), g.to(device), g.edges(etype=...) -> (src, dst)
  * dgl.batch(list) / dgl.unbatch(g)

The three semantics (src - dst, mean with zero fill, concatenating batch order) are the
build's restatement of third-party behaviour: no reference test pins them ("parity
unpinned at the DGL boundary").
"""
import contextlib

import torch

from . import function  # noqa: F401


def _canon(etype):
    if isinstance(etype, tuple):
        return etype[1] if etype[1] != 'cross' else etype
    return etype


class _DataView:
    def __init__(self, store):
        self.data = store


class _Indexer:
    def __init__(self, stores, canon=lambda k: k):
        self._stores = stores
        self._canon = canon

    def __getitem__(self, key):
        return _DataView(self._stores[self._canon(key)])


class _EdgeView(_Indexer):
    """g.edges[etype].data (dict view) and g.edges(etype=...) -> (src, dst), like DGL's EdgeView."""

    def __init__(self, g):
        super().__init__(g._edata, _canon)
        self._g = g

    def __call__(self, form='uv', order='eid', etype=None):
        return self._g._edges[_canon(etype)]


class _EdgeBatch:
    def __init__(self, src_data, dst_data, src_idx, dst_idx, edata):
        self.src = {k: v[src_idx] for k, v in src_data.items() if torch.is_tensor(v)}
        self.dst = {k: v[dst_idx] for k, v in dst_data.items() if torch.is_tensor(v)}
        self.data = edata


class HG:
    """Heterograph with node types ligand/receptor and edge types ll/rr (cross types are empty)."""

    ETYPES = {'ll': 'ligand', 'rr': 'receptor'}

    def __init__(self, num_nodes, edges, batch_nodes=None, batch_edges=None):
        self._num_nodes = dict(num_nodes)
        self._edges = {k: (s.long(), d.long()) for k, (s, d) in edges.items()}
        self._ndata = {'ligand': {}, 'receptor': {}}
        self._edata = {'ll': {}, 'rr': {}}
        self._batch_nodes = batch_nodes or {k: [v] for k, v in self._num_nodes.items()}
        self._batch_edges = batch_edges or {k: [int(s.numel())] for k, (s, d) in self._edges.items()}

    # ---- accessors -------------------------------------------------------------------------
    @property
    def nodes(self):
        return _Indexer(self._ndata)

    @property
    def edges(self):
        return _EdgeView(self)

    def all_edges(self, form='uv', order='eid', etype=None):
        return self._edges[_canon(etype)]

    def num_nodes(self, ntype):
        return self._num_nodes[ntype]

    def batch_num_nodes(self, ntype):
        return torch.tensor(self._batch_nodes[ntype], dtype=torch.int64)

    def batch_num_edges(self, etype):
        return torch.tensor(self._batch_edges[_canon(etype)], dtype=torch.int64)

    def to(self, device):
        for store in list(self._ndata.values()) + list(self._edata.values()):
            for k in list(store):
                store[k] = store[k].to(device)
        self._edges = {k: (s.to(device), d.to(device)) for k, (s, d) in self._edges.items()}
        return self

    @contextlib.contextmanager
    def local_scope(self):
        saved_n = {k: dict(v) for k, v in self._ndata.items()}
        saved_e = {k: dict(v) for k, v in self._edata.items()}
        try:
            yield
        finally:
            for k in self._ndata:
                self._ndata[k].clear()
                self._ndata[k].update(saved_n[k])
            for k in self._edata:
                self._edata[k].clear()
                self._edata[k].update(saved_e[k])

    # ---- message passing -------------------------------------------------------------------
    def apply_edges(self, func, etype):
        et = _canon(etype)
        nt = self.ETYPES[et]
        src, dst = self._edges[et]
        if isinstance(func, function._USubV):
            self._edata[et][func.out] = self._ndata[nt][func.lhs][src] - self._ndata[nt][func.rhs][dst]
        else:
            out = func(_EdgeBatch(self._ndata[nt], self._ndata[nt], src, dst, self._edata[et]))
            self._edata[et].update(out)

    def update_all(self, msg, red, etype):
        et = _canon(etype)
        nt = self.ETYPES[et]
        assert isinstance(msg, function._CopyEdge) and isinstance(red, function._Mean)
        assert msg.out == red.msg
        src, dst = self._edges[et]
        m = self._edata[et][msg.field]
        n = self._num_nodes[nt]
        acc = torch.zeros((n,) + tuple(m.shape[1:]), dtype=m.dtype, device=m.device)
        acc = acc.index_add(0, dst, m)
        deg = torch.zeros(n, dtype=m.dtype, device=m.device).index_add(
            0, dst, torch.ones(dst.numel(), dtype=m.dtype, device=m.device))
        deg = deg.clamp(min=1.0).view((n,) + (1,) * (m.dim() - 1))
        self._ndata[nt][red.out] = acc / deg


def heterograph(data_dict, num_nodes_dict):
    edges = {}
    for (s, et, d), (u, v) in data_dict.items():
        if et == 'cross':
            assert len(u) == 0
            continue
        edges[et] = (torch.as_tensor(u), torch.as_tensor(v))
    return HG(num_nodes_dict, edges)


def batch(graphs):
    num_nodes = {t: sum(g._num_nodes[t] for g in graphs) for t in ('ligand', 'receptor')}
    batch_nodes = {t: [n for g in graphs for n in g._batch_nodes[t]] for t in ('ligand', 'receptor')}
    batch_edges = {e: [n for g in graphs for n in g._batch_edges[e]] for e in ('ll', 'rr')}
    edges = {}
    for et, nt in HG.ETYPES.items():
        off = 0
        ss, dd = [], []
        for g in graphs:
            s, d = g._edges[et]
            ss.append(s + off)
            dd.append(d + off)
            off += g._num_nodes[nt]
        edges[et] = (torch.cat(ss), torch.cat(dd))
    out = HG(num_nodes, edges, batch_nodes, batch_edges)
    for nt in ('ligand', 'receptor'):
        for k in graphs[0]._ndata[nt]:
            out._ndata[nt][k] = torch.cat([g._ndata[nt][k] for g in graphs], dim=0)
    for et in ('ll', 'rr'):
        for k in graphs[0]._edata[et]:
            out._edata[et][k] = torch.cat([g._edata[et][k] for g in graphs], dim=0)
    return out


def unbatch(g):
    outs = []
    noff = {'ligand': 0, 'receptor': 0}
    eoff = {'ll': 0, 'rr': 0}
    nb = len(g._batch_nodes['ligand'])
    for i in range(nb):
        num_nodes = {t: g._batch_nodes[t][i] for t in ('ligand', 'receptor')}
        edges = {}
        for et, nt in HG.ETYPES.items():
            ne = g._batch_edges[et][i]
            s, d = g._edges[et]
            edges[et] = (s[eoff[et]:eoff[et] + ne] - noff[nt], d[eoff[et]:eoff[et] + ne] - noff[nt])
        h = HG(num_nodes, edges)
        for nt in ('ligand', 'receptor'):
            for k, v in g._ndata[nt].items():
                h._ndata[nt][k] = v[noff[nt]:noff[nt] + num_nodes[nt]]
        for et in ('ll', 'rr'):
            ne = g._batch_edges[et][i]
            for k, v in g._edata[et].items():
                h._edata[et][k] = v[eoff[et]:eoff[et] + ne]
        for nt in ('ligand', 'receptor'):
            noff[nt] += num_nodes[nt]
        for et in ('ll', 'rr'):
            eoff[et] += g._batch_edges[et][i]
        outs.append(h)
    return outs


class _HomoGraph:
    """dgl.graph(([], []), idtype=...) + add_nodes / add_edges / ndata / edata: what the reference's graph construction
    (src/utils/protein_utils.py:331-333, 363-394) uses to hold one protein's k-NN graph."""

    def __init__(self):
        self._n = 0
        self._src, self._dst = [], []
        self.ndata, self.edata = {}, {}

    def add_nodes(self, n):
        self._n += int(n)

    def add_edges(self, u, v):
        self._src.extend(int(a) for a in u)
        self._dst.extend(int(a) for a in v)

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return len(self._src)

    def edges(self):
        return torch.tensor(self._src, dtype=torch.int32), torch.tensor(self._dst, dtype=torch.int32)


def graph(data, idtype=None):
    assert len(data[0]) == 0 and len(data[1]) == 0
    return _HomoGraph()
