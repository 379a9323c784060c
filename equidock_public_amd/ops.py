"""Operator-level autograd wrappers of the C ABI (include/equidock_hip.h) and the standalone IEGMN layer built from them.

    aggr_msg, x_new = ops.edge_message(packed, P, Q, x, layer_params..., eta=..., use_dist=..., use_he=...)
    out             = ops.cross_attention(packed, q, k, v)
    x_l', h_l', x_r', h_r' = ops.layer_forward(layer, pair_graph, ...)        # IEGMN_Layer.forward's body

Inside IEGMN the layer loop never comes through here (one C call per pass, model._IEGMNFunction); these wrappers serve a
caller that wants ONE IEGMN_Layer (reference signature, src/model/rigid_docking_model.py:189-352) or a single operator with
autograd.  The three operators of a layer - the edge messages + coordinate update (:204-237, 263-292), the
block-diagonal cross attention (:46-64, 244-256) and the node update (node_mlp + skip connection, :319-337) - run in the
HIP library forward and backward (eqd_edge_message_fwd / _bwd, eqd_cross_attention_fwd / _bwd, eqd_node_update_fwd /
_bwd); the five node projections in front of them (P, Q, q, k, v: one nn.Linear each) are torch's.  No CPU fallback:
tensors must be on the GPU.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib


def _f32(t, what):
    return _lib.require_device(t.to(torch.float32).contiguous(), what)


class _GraphView:
    """EqdGraph struct of a packed batch with its own coordinate (x0) and edge-feature (he) pointers."""

    def __init__(self, packed, x0, he):
        self.packed, self.x0, self.he = packed, x0, he      # keep the tensors alive
        self.gs = _lib.graph_struct(packed)
        self.gs.x0 = x0.data_ptr()
        if he is not None:
            self.gs.he = he.data_ptr()


class _EdgeMessage(torch.autograd.Function):
    """eqd_edge_message_fwd / _bwd.  W1 is edge_mlp.0.weight [64, 2 d_in + 42]: the kernels use (and differentiate) its
    columns >= 2 d_in; the node part enters through P = h W1[:, :d]^T and Q = h W1[:, d:2d]^T + b1."""

    @staticmethod
    def forward(ctx, view, d_in, eta, slope, use_dist, use_he, P, Q, x, W1, ln_g, ln_b, W2, b2, Wc1, bc1, wc2, bc2):
        lib = _lib.load_library()
        dev = P.device
        ts = [_f32(t, 'edge_message operand') for t in (P, Q, x, W1, ln_g, ln_b, W2, b2, Wc1, bc1, wc2, bc2)]
        P_, Q_, x_, W1_, lg, lb, W2_, b2_, Wc1_, bc1_, wc2_, bc2_ = ts
        ep = _lib.EqdEdgeParams()
        ep.W1, ep.ldw1, ep.d_in = W1_.data_ptr(), W1_.shape[1], int(d_in)
        ep.ln_g, ep.ln_b, ep.W2, ep.b2 = lg.data_ptr(), lb.data_ptr(), W2_.data_ptr(), b2_.data_ptr()
        ep.Wc1, ep.bc1, ep.wc2, ep.bc2 = Wc1_.data_ptr(), bc1_.data_ptr(), wc2_.data_ptr(), bc2_.data_ptr()
        ep.slope, ep.ln_eps, ep.eta = float(slope), 1e-5, float(eta)
        ep.use_dist, ep.use_he = int(bool(use_dist)), int(bool(use_he))
        N = view.packed.n_nodes
        aggr = torch.empty(N, 64, dtype=torch.float32, device=dev)
        xnew = torch.empty(N, 3, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_edge_message_fwd(C.byref(view.gs), C.byref(ep), _lib.ptr(P_), _lib.ptr(Q_), _lib.ptr(x_),
                                                _lib.ptr(aggr), _lib.ptr(xnew), _lib.stream_ptr(dev)))
        ctx.view, ctx.ep, ctx.ts = view, ep, ts
        return aggr, xnew

    @staticmethod
    def backward(ctx, d_aggr, d_xnew):
        lib = _lib.load_library()
        view, ep = ctx.view, ctx.ep
        P_, Q_, x_, W1_, lg, lb, W2_, b2_, Wc1_, bc1_, wc2_, bc2_ = ctx.ts
        dev = P_.device
        N = view.packed.n_nodes
        z = dict(dtype=torch.float32, device=dev)
        d_aggr = torch.zeros(N, 64, **z) if d_aggr is None else _f32(d_aggr, 'd aggr_msg')
        d_xnew = torch.zeros(N, 3, **z) if d_xnew is None else _f32(d_xnew, 'd x_new')
        dP, dQ, dx = torch.empty(N, 64, **z), torch.empty(N, 64, **z), torch.empty(N, 3, **z)
        g = {k: torch.zeros_like(t) for k, t in (('W1', W1_), ('lg', lg), ('lb', lb), ('W2', W2_), ('b2', b2_),
                                                 ('Wc1', Wc1_), ('bc1', bc1_), ('wc2', wc2_), ('bc2', bc2_))}
        eg = _lib.EqdEdgeGrads()
        eg.dW1, eg.ldw1 = g['W1'].data_ptr(), W1_.shape[1]
        eg.dln_g, eg.dln_b, eg.dW2, eg.db2 = g['lg'].data_ptr(), g['lb'].data_ptr(), g['W2'].data_ptr(), g['b2'].data_ptr()
        eg.dWc1, eg.dbc1, eg.dwc2, eg.dbc2 = (g[k].data_ptr() for k in ('Wc1', 'bc1', 'wc2', 'bc2'))
        with _lib.device_guard(dev):
            wsb = lib.eqd_edge_message_bwd_workspace_bytes(C.byref(view.gs))
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.check(lib.eqd_edge_message_bwd(C.byref(view.gs), C.byref(ep), _lib.ptr(P_), _lib.ptr(Q_), _lib.ptr(x_),
                                                _lib.ptr(d_aggr), _lib.ptr(d_xnew), _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dx),
                                                C.byref(eg), _lib.ptr(ws), C.c_size_t(wsb), _lib.stream_ptr(dev)))
        return (None, None, None, None, None, None, dP, dQ, dx, g['W1'], g['lg'], g['lb'], g['W2'], g['b2'], g['Wc1'],
                g['bc1'], g['wc2'], g['bc2'])


class _CrossAttention(torch.autograd.Function):
    """eqd_cross_attention_fwd / _bwd: out_i = sum_j softmax_j(q_i . k_j) v_j over the partner protein of the same pair."""

    @staticmethod
    def forward(ctx, view, q, k, v):
        lib = _lib.load_library()
        dev = q.device
        q_, k_, v_ = (_f32(t, 'attention operand') for t in (q, k, v))
        d = q_.shape[1]
        out = torch.empty_like(q_)
        lse = torch.empty(q_.shape[0], dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_cross_attention_fwd(C.byref(view.gs), int(d), _lib.ptr(q_), _lib.ptr(k_), _lib.ptr(v_),
                                                   _lib.ptr(out), _lib.ptr(lse), _lib.stream_ptr(dev)))
        ctx.view = view
        ctx.save_for_backward(q_, k_, v_, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load_library()
        q_, k_, v_, out, lse = ctx.saved_tensors
        dev = q_.device
        d_out = _f32(d_out, 'd attention output')
        dq, dk, dv = torch.empty_like(q_), torch.empty_like(q_), torch.empty_like(q_)
        delta = torch.empty_like(lse)
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_cross_attention_bwd(C.byref(ctx.view.gs), int(q_.shape[1]), _lib.ptr(q_), _lib.ptr(k_),
                                                   _lib.ptr(v_), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(d_out), _lib.ptr(dq),
                                                   _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(delta), _lib.stream_ptr(dev)))
        return None, dq, dk, dv


class _NodeUpdate(torch.autograd.Function):
    """eqd_node_update_fwd / _bwd: node_mlp([h | aggr_msg | aggr_cross | h0]) and the skip connection
    (rigid_docking_model.py:319-337).  aggr_cross rows are ld_cross = aggr_cross.shape[1] floats wide (>= d_in: the attention
    operator's padded rows of a 69-wide layer are taken as they are)."""

    @staticmethod
    def forward(ctx, skip_weight_h, slope, ln_eps, h, aggr_msg, aggr_cross, h0, Wn1, bn1, ln_g, ln_b, Wn2, bn2):
        lib = _lib.load_library()
        dev = h.device
        ts = [_f32(t, 'node_update operand') for t in (h, aggr_msg, aggr_cross, h0, Wn1, bn1, ln_g, ln_b, Wn2, bn2)]
        h_, am, ac, h0_, Wn1_, bn1_, lg, lb, Wn2_, bn2_ = ts
        rows, d = h_.shape
        p = _lib.EqdNodeUpdateParams()
        p.d_in, p.d0, p.d_out, p.ld_cross = int(d), int(h0_.shape[1]), int(Wn2_.shape[0]), int(ac.shape[1])
        p.Wn1, p.bn1, p.ln_g, p.ln_b, p.Wn2, p.bn2 = (t.data_ptr() for t in (Wn1_, bn1_, lg, lb, Wn2_, bn2_))
        p.skip_weight_h, p.slope, p.ln_eps, p.bf16, p.drop_mul = float(skip_weight_h), float(slope), float(ln_eps), 0, None
        if Wn1_.shape[1] != p.d0 + 2 * d + 64 or am.shape[1] != 64:
            raise _lib.EquidockHipError(f"node_update: node_mlp.0.weight is {tuple(Wn1_.shape)}, expected [{d}, {p.d0 + 2 * d + 64}]")
        z = dict(dtype=torch.float32, device=dev)
        h_out, y_act, a1n = torch.empty(rows, p.d_out, **z), torch.empty(rows, d, **z), torch.empty(rows, d, **z)
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_node_update_fwd(int(rows), C.byref(p), _lib.ptr(h_), _lib.ptr(am), _lib.ptr(ac), _lib.ptr(h0_),
                                               _lib.ptr(h_out), _lib.ptr(y_act), _lib.ptr(a1n), _lib.stream_ptr(dev)))
        # inputs, parameters and the saved activations go through save_for_backward (autograd's in-place version checks; no
        # reference cycle through ctx); the pointers of EqdNodeUpdateParams are rebuilt from them in backward
        ctx.scalars = (float(skip_weight_h), float(slope), float(ln_eps))
        ctx.save_for_backward(*ts, y_act, a1n)
        return h_out

    @staticmethod
    def backward(ctx, d_h_out):
        lib = _lib.load_library()
        h_, am, ac, h0_, Wn1_, bn1_, lg, lb, Wn2_, bn2_, y_act, a1n = ctx.saved_tensors
        dev = h_.device
        rows, d = h_.shape
        p = _lib.EqdNodeUpdateParams()
        p.d_in, p.d0, p.d_out, p.ld_cross = int(d), int(h0_.shape[1]), int(Wn2_.shape[0]), int(ac.shape[1])
        p.Wn1, p.bn1, p.ln_g, p.ln_b, p.Wn2, p.bn2 = (t.data_ptr() for t in (Wn1_, bn1_, lg, lb, Wn2_, bn2_))
        p.skip_weight_h, p.slope, p.ln_eps = ctx.scalars
        p.bf16, p.drop_mul = 0, None
        d_h_out = _f32(d_h_out, 'd h_out')
        z = dict(dtype=torch.float32, device=dev)
        d_h, d_am, d_h0 = torch.empty_like(h_), torch.empty_like(am), torch.empty_like(h0_)
        d_ac = torch.zeros_like(ac)      # (padding columns beyond d_in stay zero)
        g = [torch.zeros_like(t) for t in (Wn1_, bn1_, lg, lb, Wn2_, bn2_)]
        gr = _lib.EqdNodeUpdateGrads()
        gr.dWn1, gr.dbn1, gr.dln_g, gr.dln_b, gr.dWn2, gr.dbn2 = (t.data_ptr() for t in g)
        with _lib.device_guard(dev):
            wsb = lib.eqd_node_update_bwd_workspace_bytes(int(rows), C.byref(p))
            ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
            _lib.check(lib.eqd_node_update_bwd(int(rows), C.byref(p), _lib.ptr(h_), _lib.ptr(am), _lib.ptr(ac), _lib.ptr(h0_),
                                               _lib.ptr(y_act), _lib.ptr(a1n), _lib.ptr(d_h_out), _lib.ptr(d_h), _lib.ptr(d_am),
                                               _lib.ptr(d_ac), _lib.ptr(d_h0), C.byref(gr), _lib.ptr(ws), C.c_size_t(wsb),
                                               _lib.stream_ptr(dev)))
        return (None, None, None, d_h, d_am, d_ac, d_h0) + tuple(g)


def graph_view(packed, x0=None, he=None):
    """Kernel-side view of a packed batch (graph.PackedGraph) with the given original coordinates [n_nodes, 3] (ligand rows
    first; default: the batch's own) and edge features [n_edges, 27] in the PACKED edge order (default: the batch's own)."""
    x0 = packed.x0 if x0 is None else _f32(x0, 'x0')
    return _GraphView(packed, x0, None if he is None else _f32(he, 'he'))


def edge_message(view, P, Q, x, W1, ln_g, ln_b, W2, b2, Wc1, bc1, wc2, bc2, d_in, eta=0.0, slope=0.01, use_dist=True,
                 use_he=True):
    """(aggr_msg [n_nodes, 64], x_new [n_nodes, 3]) = eqd_edge_message_fwd, differentiable w.r.t. P, Q, x and the nine
    parameter tensors (W1: only its columns >= 2 d_in are used here)."""
    return _EdgeMessage.apply(view, d_in, eta, slope, use_dist, use_he, P, Q, x, W1, ln_g, ln_b, W2, b2, Wc1, bc1, wc2, bc2)


def cross_attention(view, q, k, v):
    """Block-diagonal cross attention of both directions, [n_nodes, d] each (ligand rows first), differentiable."""
    return _CrossAttention.apply(view, q, k, v)


def node_update(h, aggr_msg, aggr_cross, h0, Wn1, bn1, ln_g, ln_b, Wn2, bn2, skip_weight_h, slope=0.01, ln_eps=1e-5):
    """h' [rows, 64] = eqd_node_update_fwd: skip(node_mlp([h | aggr_msg | aggr_cross | h0])), differentiable w.r.t. the four
    inputs and the six parameter tensors (eqd_node_update_bwd)."""
    return _NodeUpdate.apply(skip_weight_h, slope, ln_eps, h, aggr_msg, aggr_cross, h0, Wn1, bn1, ln_g, ln_b, Wn2, bn2)


def layer_supported(layer):
    """The standalone layer runs its two heavy operators in the HIP library for the published configuration family
    (model.hip_path_supported's conditions on one layer) when no dropout mask is due."""
    em, cm = layer.edge_mlp, layer.coors_mlp
    return (not layer.fine_tune and isinstance(em[2], torch.nn.LeakyReLU) and isinstance(em[3], torch.nn.LayerNorm)
            and isinstance(cm[3], torch.nn.Identity) and isinstance(layer.final_h_layernorm_layer, torch.nn.Identity)
            and isinstance(layer.node_mlp[2], torch.nn.LeakyReLU) and isinstance(layer.node_mlp[3], torch.nn.LayerNorm)
            and isinstance(layer.node_norm, torch.nn.Identity)
            and layer.out_feats_dim == 64 and em[0].in_features == 2 * layer.h_feats_dim + 42
            and not (layer.training and layer.dropout_p > 0))


def layer_forward(layer, g, x_l, h_l, h0_l, he_l, x0_l, x_r, h_r, h0_r, he_r, x0_r):
    """IEGMN_Layer.forward (rigid_docking_model.py:189-352) -> x_ligand', h_ligand', x_receptor', h_receptor'.
    `g`: PairGraph batch.  Differentiable w.r.t. the coordinates, the node features and every parameter; the edge features
    are constants (graph data), as inside the model."""
    packed = g.pack()
    nl = x_l.shape[0]
    d = layer.h_feats_dim
    dev = x_l.device
    x = torch.cat([x_l, x_r], 0)
    h = torch.cat([h_l, h_r], 0)
    h0 = torch.cat([h0_l, h0_r], 0)
    x0 = torch.cat([x0_l, x0_r], 0)
    he_raw = torch.cat([he_l, he_r], 0).detach()
    he = he_raw[packed.edge_perm.to(dev).long()] if he_raw.numel() else he_raw      # packed (destination-sorted) edge order
    view = graph_view(packed, x0.detach(), he)
    W1, b1 = layer.edge_mlp[0].weight, layer.edge_mlp[0].bias
    P = F.linear(h, W1[:, :d])                                                   # :226-237, first Linear split by operand
    Q = F.linear(h, W1[:, d:2 * d], b1)
    slope = layer.edge_mlp[2].negative_slope
    # the library's x_new = eta x0 + (1 - eta) x + x_update with x0 a constant of the graph view: called with eta = 0 and
    # completed here, so that autograd also reaches the original coordinates
    aggr_msg, x_upd = edge_message(view, P, Q, x, W1, layer.edge_mlp[3].weight, layer.edge_mlp[3].bias,
                                   layer.edge_mlp[4].weight, layer.edge_mlp[4].bias, layer.coors_mlp[0].weight,
                                   layer.coors_mlp[0].bias, layer.coors_mlp[4].weight, layer.coors_mlp[4].bias, d_in=d,
                                   eta=0.0, slope=slope, use_dist=layer.use_dist_in_layers, use_he=True)
    eta = layer.x_connection_init
    x_new = x_upd + eta * (x0 - x)                                                # :286-292
    if layer.cross_msgs:                                                          # :244-256
        cross = cross_attention(view, layer.att_mlp_Q(h), layer.att_mlp_K(h), layer.att_mlp_V(h))
    else:
        cross = torch.zeros_like(h)
    nm = layer.node_mlp                                                           # :319-337, eqd_node_update_fwd / _bwd
    upd = node_update(h, aggr_msg, cross, h0, nm[0].weight, nm[0].bias, nm[3].weight, nm[3].bias, nm[4].weight, nm[4].bias,
                      layer.skip_weight_h, slope=nm[2].negative_slope, ln_eps=nm[3].eps)
    return x_new[:nl], upd[:nl], x_new[nl:], upd[nl:]
