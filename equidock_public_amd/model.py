"""Drop-in nn.Modules for the reference's IEGMN hot path, running on MI355X through
libequidock_hip.so.

Same operator surface as src/model/rigid_docking_model.py in the reference:

    Rigid_Body_Docking_Net(args, log=None)                     (:611-696)
    IEGMN(args, n_lays, fine_tune, log=None)                   (:360-606)
    IEGMN_Layer(orig_h_feats_dim, h_feats_dim, out_feats_dim, fine_tune, args, log=None)   (:82-356)
    model(batch_hetero_graph, epoch) -> (ligand_coords_list, keypts_ligand_list,
                                         keypts_receptor_list, rotation_list, translation_list)

with identical constructor arguments, `args` keys, parameter names/shapes (so
`load_state_dict(checkpoint['state_dict'])` of a reference checkpoint works, :109 of
src/inference_rigid.py) and default initialisation (the parameter-holding sub-modules are the
same torch modules created in the same order, so a given torch seed yields the reference's
initial weights).  The arithmetic does NOT run in those sub-modules: forward and backward are
two C calls into the HIP library (include/equidock_hip.h) wrapped in one autograd.Function.

The published configuration family (src/utils/args.py:227-280: nonlin 'lkyrelu', layer_norm 'LN',
layer_norm_coors '0', final_h_layer_norm '0', hidden/embedding width 64, no fine-tune stage) runs in the HIP
library - in eval and in training mode, with or without dropout: nn.Dropout's masks are drawn with torch's generator in
the reference's consumption order (DropoutMasks) and applied inside the kernels - and there is no CPU fallback for it: a
missing library or a CPU tensor raises.  The reference's other options - swish, BatchNorm / LayerNorm placements,
GraphNorm (src/utils/graph_norm.py), the fine-tune stage - build the same sub-modules and run them through torch
operators on the tensors' device (equidock_public_amd/torch_path.py), so that every configuration of the reference
constructs, loads its checkpoint and trains.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib
from .graph import PairGraph


def get_non_lin(type, negative_slope):
    """rigid_docking_model.py:10-15."""
    if type == 'swish':
        return nn.SiLU()
    if type != 'lkyrelu':
        raise ValueError(f"nonlin='{type}': the reference knows 'swish' and 'lkyrelu'")
    return nn.LeakyReLU(negative_slope=negative_slope)


def get_layer_norm(layer_norm_type, dim):
    """rigid_docking_model.py:18-24."""
    if layer_norm_type == 'BN':
        return nn.BatchNorm1d(dim)
    if layer_norm_type == 'LN':
        return nn.LayerNorm(dim)
    return nn.Identity()


class GraphNorm(nn.Module):
    """src/utils/graph_norm.py:7-41: per-graph (x - mean) / (std + eps) with the UNBIASED std and eps added to the std,
    then an affine map.  `counts`: nodes per graph of the batch (the reference reads them from g.batch_num_nodes)."""

    def __init__(self, num_features, eps=1e-5, affine=True, is_node=True):
        super().__init__()
        self.eps, self.num_features, self.affine, self.is_node = eps, num_features, affine, is_node
        if affine:
            self.gamma = nn.Parameter(torch.ones(num_features))
            self.beta = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter('gamma', None)
            self.register_parameter('beta', None)

    def forward(self, counts, h):
        parts = [(x - x.mean(dim=0, keepdim=True)) / (x.std(dim=0, keepdim=True) + self.eps)
                 for x in torch.split(h, [int(c) for c in counts])]
        out = torch.cat(parts, 0)
        return self.gamma * out + self.beta if self.affine else out


def get_final_h_layer_norm(layer_norm_type, dim):
    """rigid_docking_model.py:27-36."""
    if layer_norm_type == 'BN':
        return nn.BatchNorm1d(dim)
    if layer_norm_type == 'LN':
        return nn.LayerNorm(dim)
    if layer_norm_type == 'GN':
        return GraphNorm(dim)
    if layer_norm_type != '0':
        raise ValueError(f"final_h_layer_norm='{layer_norm_type}': the reference knows BN, LN, GN and '0'")
    return nn.Identity()


def hip_path_supported(args, fine_tune=False):
    """True when the configuration is the published family (src/utils/args.py:227-280), which runs as two C calls into
    the HIP library (any dropout, training or eval mode); every other reference option - swish, BatchNorm / LayerNorm
    placements, GraphNorm, the fine-tune stage - runs the same modules through torch operators on the GPU
    (equidock_public_amd/torch_path.py)."""
    return (not fine_tune and args['nonlin'] == 'lkyrelu' and args['layer_norm'] == 'LN' and args['layer_norm_coors'] == '0'
            and args['final_h_layer_norm'] == '0' and args['iegmn_lay_hid_dim'] == 64 and args['residue_emb_dim'] <= 64
            and args['input_edge_feats_dim'] == 27)


class IEGMN_Layer(nn.Module):
    """Parameter container of one IEGMN layer (same names/shapes as the reference, :119-159)."""

    def __init__(self, orig_h_feats_dim, h_feats_dim, out_feats_dim, fine_tune, args, log=None):
        super().__init__()
        input_edge_feats_dim = args['input_edge_feats_dim']
        dropout = args['dropout']
        nonlin = args['nonlin']
        slope = args['leakyrelu_neg_slope']
        self.cross_msgs = args['cross_msgs']
        self.use_dist_in_layers = args['use_dist_in_layers']
        self.skip_weight_h = args['skip_weight_h']
        self.x_connection_init = args['x_connection_init']
        self.fine_tune = fine_tune
        self.final_h_layer_norm = args['final_h_layer_norm']
        self.debug = args['debug']
        self.log = log
        self.h_feats_dim = h_feats_dim
        self.out_feats_dim = out_feats_dim
        self.dropout_p = dropout
        self.all_sigmas_dist = [1.5 ** x for x in range(15)]

        self.edge_mlp = nn.Sequential(
            nn.Linear((h_feats_dim * 2) + input_edge_feats_dim + len(self.all_sigmas_dist), out_feats_dim),
            nn.Dropout(dropout), get_non_lin(nonlin, slope), get_layer_norm(args['layer_norm'], out_feats_dim),
            nn.Linear(out_feats_dim, out_feats_dim))
        self.node_norm = nn.Identity()
        self.att_mlp_Q = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim, bias=False), get_non_lin(nonlin, slope))
        self.att_mlp_K = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim, bias=False), get_non_lin(nonlin, slope))
        self.att_mlp_V = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim, bias=False))
        self.node_mlp = nn.Sequential(
            nn.Linear(orig_h_feats_dim + 2 * h_feats_dim + out_feats_dim, h_feats_dim), nn.Dropout(dropout),
            get_non_lin(nonlin, slope), get_layer_norm(args['layer_norm'], h_feats_dim),
            nn.Linear(h_feats_dim, out_feats_dim))
        self.final_h_layernorm_layer = get_final_h_layer_norm(self.final_h_layer_norm, out_feats_dim)
        self.coors_mlp = nn.Sequential(
            nn.Linear(out_feats_dim, out_feats_dim), nn.Dropout(dropout), get_non_lin(nonlin, slope),
            get_layer_norm(args['layer_norm_coors'], out_feats_dim), nn.Linear(out_feats_dim, 1))
        if self.fine_tune:      # rigid_docking_model.py:161-174
            self.att_mlp_cross_coors_Q = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim, bias=False),
                                                       get_non_lin(nonlin, slope))
            self.att_mlp_cross_coors_K = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim, bias=False),
                                                       get_non_lin(nonlin, slope))
            self.att_mlp_cross_coors_V = nn.Sequential(nn.Linear(h_feats_dim, h_feats_dim), get_non_lin(nonlin, slope),
                                                       nn.Linear(h_feats_dim, 1))

    def param_table(self):
        """The 19 tensors in the order of include/equidock_hip.h's parameter table."""
        return [self.edge_mlp[0].weight, self.edge_mlp[0].bias, self.edge_mlp[3].weight, self.edge_mlp[3].bias,
                self.edge_mlp[4].weight, self.edge_mlp[4].bias, self.att_mlp_Q[0].weight, self.att_mlp_K[0].weight,
                self.att_mlp_V[0].weight, self.node_mlp[0].weight, self.node_mlp[0].bias, self.node_mlp[3].weight,
                self.node_mlp[3].bias, self.node_mlp[4].weight, self.node_mlp[4].bias, self.coors_mlp[0].weight,
                self.coors_mlp[0].bias, self.coors_mlp[4].weight, self.coors_mlp[4].bias]

    def forward(self, hetero_graph, coors_ligand, h_feats_ligand, original_ligand_node_features,
                original_edge_feats_ligand, orig_coors_ligand, coors_receptor, h_feats_receptor,
                original_receptor_node_features, original_edge_feats_receptor, orig_coors_receptor):
        """One layer on its own, with the reference's signature (rigid_docking_model.py:189-352).  Inside IEGMN the layer
        loop of the published configuration never comes through here - it runs as one C call (eqd_model_forward).  A
        layer called by itself runs its two heavy operators - edge messages + coordinate update, cross attention - in the
        HIP library, forward and backward (equidock_public_amd/ops.py: eqd_edge_message_*, eqd_cross_attention_*), and its
        node-level Linears through its own torch sub-modules, for the published configuration; the reference's other
        options (and dropout > 0 while training) compose everything from torch operators (torch_path.py).
        Differentiable w.r.t. coordinates, node features and parameters either way."""
        from . import ops, torch_path
        from .graph import from_dgl
        fn = ops.layer_forward if ops.layer_supported(self) else torch_path.layer_forward
        return fn(self, from_dgl(hetero_graph), coors_ligand, h_feats_ligand,
                  original_ligand_node_features, original_edge_feats_ligand, orig_coors_ligand,
                  coors_receptor, h_feats_receptor, original_receptor_node_features,
                  original_edge_feats_receptor, orig_coors_receptor)

    def __repr__(self):
        return f"IEGMN Layer (HIP) h_feats_dim={self.h_feats_dim} out_feats_dim={self.out_feats_dim}"


def _dgl_signature(g):
    """Identity of the tensors a DGL heterograph currently holds for the accessors from_dgl() reads (None when the object
    does not look like one: from_dgl then raises its own error)."""
    try:
        ts = [g.nodes['ligand'].data[k] for k in ('res_feat', 'x', 'new_x', 'mu_r_norm')]
        ts += [g.nodes['receptor'].data[k] for k in ('res_feat', 'x', 'mu_r_norm')]
        ts += [g.edges[c].data['he'] for c in (('ligand', 'll', 'ligand'), ('receptor', 'rr', 'receptor'))]
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in ts)
    except (KeyError, AttributeError, TypeError):
        return None


def adapt_graph(batch_hetero_graph):
    """The PairGraph of whatever the caller handed over.  The reference's callers pass a batched DGL heterograph
    (src/train.py:94-100): it is adapted (duck-typed, tensors shared) and the adaptation is remembered on the object so that
    its packed layout is built once - as long as the graph still holds the SAME tensors: DGL replaces a tensor when the user
    assigns node / edge data (`g.nodes['ligand'].data['new_x'] = ...`, what the reference's augmentation and fine-tune stage
    do), and an in-place write bumps its version, so the cache is keyed on (data_ptr, version, shape) of every tensor it
    adapted.  Shared by Rigid_Body_Docking_Net.forward_batched and IEGMN.run, so that `model(batched_dgl_graph, epoch)` -
    the reference's call - re-uses the adaptation and the PairGraph-side caches (packed layout, saved state, dropout ones)
    from step to step as well."""
    if isinstance(batch_hetero_graph, PairGraph):
        return batch_hetero_graph
    sig = _dgl_signature(batch_hetero_graph)
    cached = getattr(batch_hetero_graph, '_eqd_pair_graph', None)
    if cached is None or sig is None or cached[0] != sig:
        from .graph import from_dgl
        cached = (sig, from_dgl(batch_hetero_graph))
        try:
            batch_hetero_graph._eqd_pair_graph = cached
        except AttributeError:
            pass
    return cached[1]


def flat_layout(tensors):
    """Offsets (in floats, 64-float aligned) of each unique parameter in a flat gradient buffer."""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() + 63) // 64 * 64
    return offs, total


class DropoutMasks:
    """The nn.Dropout keep masks of ONE training-mode forward (include/equidock_hip.h: EqdDropout).

    Drawn with torch's own dropout on tensors of the reference's shapes, in the reference's consumption order - per layer
    edge_mlp.1 on the ll then the rr edges, coors_mlp.1 ll / rr, node_mlp.1 ligand / receptor
    (rigid_docking_model.py:236-237, 263-265, 319-337), then mlp_h_mean_ROT.1 per pair, receptor before ligand (:524-529) -
    so that for a given torch seed the masks are the ones the reference's nn.Dropout modules would draw on the same
    device (a mask depends on the generator state and the tensor's shape, not on its values).  `device`: where the draws
    happen - the tensors' device by default; 'cpu' reproduces a CPU run of the reference (that is how the recorded
    `dropout_train` vectors of tests/golden/variants.npz are matched on a GPU).  The masks are then re-ordered to the
    library's edge order and bit-packed (edges) / kept as 0 | 1/(1-p) factors (nodes)."""

    def __init__(self, p, edge_z1, edge_ch, node, head):
        self.p, self.edge_z1, self.edge_ch, self.node, self.head = float(p), edge_z1, edge_ch, node, head

    def c_struct(self):
        d = _lib.EqdDropout()
        d.p = self.p
        d.edge_z1, d.edge_ch = self.edge_z1.data_ptr(), self.edge_ch.data_ptr()
        d.node, d.head = self.node.data_ptr(), self.head.data_ptr()
        return d

    @staticmethod
    def draw_library(iegmn, packed):
        """args['hip_dropout_masks'] = 'library' (not a reference option): the masks of one forward from ONE launch of
        eqd_dropout_draw - counter-based (Philox4x32-10), keyed by a 64-bit word drawn on the device from torch's generator
        (so torch.manual_seed still fixes the run, and a replayed hipGraph of the step sees fresh masks every replay).
        Same Bernoulli law as nn.Dropout, not torch's random stream; no [E, 64] tensors are materialised (the 'torch'
        source moves ~9 GB per step through HBM for them at 64 x (300, 300)).  The default since round 4."""
        p = float(iegmn.args['dropout'])
        dev = packed.x0.device
        desc, gs = iegmn._desc(), packed.c_struct()
        L, E, N = iegmn.n_lays, packed.n_edges, packed.n_nodes
        d0 = iegmn.args['residue_emb_dim'] + (5 if iegmn.use_mean_node_features else 0)
        dh = iegmn.args['iegmn_lay_hid_dim']
        seed = torch.randint(-2 ** 63, 2 ** 63 - 1, (1,), dtype=torch.int64, device=dev)
        edge_z1 = torch.empty(L, E, 2, dtype=torch.int32, device=dev)
        edge_ch = torch.empty(L, E, 2, dtype=torch.int32, device=dev)
        node = torch.empty(N * d0 + (L - 1) * N * dh, dtype=torch.float32, device=dev)
        head = torch.empty(N, 64, dtype=torch.float32, device=dev)
        lib = _lib.load_library()
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_dropout_draw(C.byref(desc), C.byref(gs), C.c_float(p), _lib.ptr(seed), _lib.ptr(edge_z1),
                                            _lib.ptr(edge_ch), _lib.ptr(node), _lib.ptr(head), _lib.stream_ptr(dev)))
        out = DropoutMasks(p, edge_z1, edge_ch, node, head)
        out.seed = seed
        return out

    _told_default = False

    @staticmethod
    def draw(iegmn, g, packed):
        import torch.nn.functional as F
        src = iegmn.args.get('hip_dropout_masks', DEFAULT_DROPOUT_MASKS)
        if 'hip_dropout_masks' not in iegmn.args and not DropoutMasks._told_default:
            # once per process: a run compared against a reference run with the same seed needs nn.Dropout's own stream
            DropoutMasks._told_default = True
            import warnings
            warnings.warn("equidock_public_amd: training-mode dropout masks are drawn by the library (eqd_dropout_draw: same "
                          "Bernoulli law as nn.Dropout, NOT torch's random stream).  Set args['hip_dropout_masks'] = 'torch' to "
                          "consume nn.Dropout's stream in the reference's order (bit-for-bit masks of a reference run with the "
                          "same seed), or 'library' to silence this note.", stacklevel=2)
        if src == 'library':
            return DropoutMasks.draw_library(iegmn, packed)
        if src != 'torch':
            raise NotImplementedError(f"hip_dropout_masks={src!r}: 'torch' (nn.Dropout's stream) or 'library' (eqd_dropout_draw)")
        p = float(iegmn.args['dropout'])
        dev = packed.x0.device
        mdev = torch.device(getattr(iegmn, 'dropout_mask_device', None) or dev)
        lc, rc = g._batch_nodes['ligand'], g._batch_nodes['receptor']
        nl, nr = sum(lc), sum(rc)
        e_ll, e_rr = int(g._edges['ll'][0].numel()), int(g._edges['rr'][0].numel())
        d0 = iegmn.args['residue_emb_dim'] + (5 if iegmn.use_mean_node_features else 0)
        dh = iegmn.args['iegmn_lay_hid_dim']

        # one tensor of ones serves every draw (a mask depends on the generator state and the SHAPE only)
        n_ones = max(e_ll, e_rr, 1) * 64
        n_ones = max(n_ones, max(nl, nr, 1) * max(d0, dh))
        ones = getattr(packed, '_dropout_ones', None)
        if ones is None or ones.device != mdev or ones.numel() < n_ones:
            ones = packed._dropout_ones = torch.ones(n_ones, dtype=torch.float32, device=mdev)

        def factors(rows, width):        # what nn.Dropout multiplies a [rows, width] activation by: 0 or 1 / (1 - p)
            return F.dropout(ones[:rows * width].view(rows, width), p, True)
        E, L = e_ll + e_rr, iegmn.n_lays
        on_dev = mdev == dev and getattr(iegmn, 'dropout_pack', 'kernel') == 'kernel'
        if on_dev:                       # packed straight into the library's edge order by eqd_dropout_pack_edges
            edge_z1 = torch.empty(L, E, 2, dtype=torch.int32, device=dev)
            edge_ch = torch.empty(L, E, 2, dtype=torch.int32, device=dev)
            lib = _lib.load_library()
            stream = _lib.stream_ptr(dev)
            perm = packed.edge_perm
            if perm.dtype != torch.int64 or not perm.is_contiguous() or perm.numel() != E or perm.device != dev:
                # k_dropout_pack reads E int64 entries through the raw pointer: anything else would be an out-of-bounds read
                raise _lib.EquidockHipError(f"edge_perm must be a contiguous int64 tensor of {E} entries on {dev} "
                                            f"(got {perm.dtype}, {tuple(perm.shape)}, {perm.device})")

            def pack(fl, fr, out):
                with _lib.device_guard(dev):
                    _lib.check(lib.eqd_dropout_pack_edges(E, e_ll, _lib.ptr(fl), _lib.ptr(fr), _lib.ptr(perm), _lib.ptr(out),
                                                          stream))
        else:                            # the same packing in torch operators (any device; what the tests compare against)
            weights = (2 ** torch.arange(32, dtype=torch.int64, device=mdev)).view(1, 1, 32)
            ez, ec = [], []

            def pack(fl, fr, out):       # two [E, 64] factor tensors (ll, rr edges) -> [E_ll + E_rr, 2] packed words
                keep = (torch.cat([fl, fr], 0) > 0).view(-1, 2, 32).to(torch.int64)
                out.append((keep * weights).sum(-1).to(torch.int32))      # (int64 -> int32 wraps: the uint32's bit pattern)
        nodes = []
        for l in range(L):
            d = d0 if l == 0 else dh
            zl, zr = factors(e_ll, 64), factors(e_rr, 64)            # edge_mlp.1: ligand edges first
            pack(zl, zr, edge_z1[l] if on_dev else ez)
            del zl, zr
            cl, cr = factors(e_ll, 64), factors(e_rr, 64)            # coors_mlp.1
            pack(cl, cr, edge_ch[l] if on_dev else ec)
            del cl, cr
            nodes += [factors(nl, d).reshape(-1), factors(nr, d).reshape(-1)]      # node_mlp.1
        head_l, head_r = [], []
        for a, b in zip(lc, rc):                                       # per pair: receptor rows, then ligand rows
            head_r.append(factors(b, 64))
            head_l.append(factors(a, 64))
        head = torch.cat(head_l + head_r, 0)
        if not on_dev:
            perm = packed.edge_perm.to(mdev).long()                    # packed edge i = raw edge perm[i]
            edge_z1 = torch.stack([t[perm] for t in ez]).to(dev).contiguous()
            edge_ch = torch.stack([t[perm] for t in ec]).to(dev).contiguous()
        return DropoutMasks(p, edge_z1, edge_ch, torch.cat(nodes).to(dev).contiguous(), head.to(dev).contiguous())


# Source of the nn.Dropout keep masks in training mode (args['hip_dropout_masks']).  'library' (default since round 4):
# eqd_dropout_draw - one launch, counter-based, keyed by a word torch's generator draws on the device, so torch.manual_seed
# still fixes a run; the same Bernoulli(1 - p) law as nn.Dropout but NOT torch's random stream.  'torch': nn.Dropout's own
# stream (torch's dropout on [E, 64] tensors of ones, in the reference's consumption order) - what a bit-for-bit comparison
# with a reference run on the same device and seed needs, at +22 % of a step at 8 x (200, 200) and +17 % at 64 x (300, 300).
DEFAULT_DROPOUT_MASKS = 'library'

POISON_WORKSPACES = False      # tests: hand the library workspaces full of NaN bit patterns instead of whatever torch.empty holds


def _workspace(nbytes, dev):
    """Caller-owned scratch / saved-state buffer for the C calls.  Nothing may depend on its contents: with POISON_WORKSPACES
    (tests) every byte is 0xFF, so that a padding column or partial buffer the kernels forgot to write shows up as NaN."""
    if POISON_WORKSPACES:
        return torch.full((nbytes,), 255, dtype=torch.uint8, device=dev)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


class _IEGMNFunction(torch.autograd.Function):
    """forward = eqd_model_forward, backward = eqd_model_backward (one C call each)."""

    @staticmethod
    def forward(ctx, packed, desc, table_idx, svd_draws, need_grad, flat_state, drop, *uniq):
        # Flat-gradient mode: `uniq` is (parameter list, anchor).  The ~160 parameters are then NOT autograd inputs
        # (apply() only tracks top-level tensors): their gradients are accumulated by the C call straight into the
        # flat buffer, and the 0-d `anchor` is the one input that makes autograd call backward().  Per-parameter
        # inputs cost ~0.5 ms of host time per step in torch's Function machinery, as much as all kernel launches.
        if flat_state is not None:
            uniq = uniq[0]
        ctx.set_materialize_grads(False)
        lib = _lib.load_library()
        dev = packed.x0.device
        gs = packed.c_struct()
        tensors = uniq                      # validated (device, fp32, contiguous) when the table was cached
        dptr = [t.data_ptr() for t in uniq]
        ptrs = (C.c_void_p * len(table_idx))(*[dptr[i] for i in table_idx])
        B, K = packed.n_pairs, desc.n_heads
        f32 = dict(dtype=torch.float32, device=dev)
        lig = torch.empty(packed.n_lig, 3, **f32)
        Yl = torch.empty(B, K, 3, **f32)
        Yr = torch.empty(B, K, 3, **f32)
        T = torch.empty(B, 3, 3, **f32)
        b = torch.empty(B, 3, **f32)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        # (the scratch size also depends on the arithmetic mode and on the EQD_* switches that select the dS hand-off form of the
        #  attention backward: the mode is part of the key, _lib.reload_tunables() bumps the generation)
        key = (desc.n_layers, desc.n_heads, desc.d_emb, desc.use_mean_node_features, desc.storage_bf16, desc.cross_msgs,
               _lib.tunables_generation)
        if packed.ws_sizes.get(key) is None:
            with _lib.device_guard(dev):    # workgroup counts (hence partial-sum workspaces) follow the device's CU count
                sb = lib.eqd_model_saved_bytes(C.byref(desc), C.byref(gs))
                wb = lib.eqd_model_scratch_bytes(C.byref(desc), C.byref(gs))
            if sb == 0 or wb == 0:
                _lib.check(lib.eqd_model_check(C.byref(desc), C.byref(gs)))
            packed.ws_sizes[key] = (sb, wb)
        sb, wb = packed.ws_sizes[key]
        # the forward carves its state from `saved` when a backward will follow and from `scratch` otherwise: only one
        # of the two is ever touched (the state is hundreds of MB at 64 x (300, 300))
        saved = _workspace(sb, dev) if need_grad else None
        # (bf16 storage mode: a forward that saves state also needs the scratch workspace - the fp32 tensors the forward itself
        #  reads but the backward only needs rounded to bf16 are transients there; the same buffer then serves the backward)
        fwd_scratch = (not need_grad) or bool(desc.storage_bf16)
        scratch = _workspace(wb, dev) if fwd_scratch else None
        if svd_draws is not None:
            svd_draws = _lib.require_device(svd_draws.to(torch.float32).contiguous(), 'svd_draws')
        if _lib.profiling:
            lib.eqd_profile_mark(b'(before forward: zero-grad fill, allocations)')
        dstruct = None if drop is None else drop.c_struct()
        with _lib.device_guard(dev):        # kernels and memsets go to the tensors' device, whatever the current one is
            _lib.check(lib.eqd_model_forward(
                C.byref(desc), C.byref(gs), ptrs, None if dstruct is None else C.byref(dstruct),
                _lib.ptr(svd_draws), _lib.ptr(lig), _lib.ptr(Yl), _lib.ptr(Yr),
                _lib.ptr(T), _lib.ptr(b), _lib.ptr(status), _lib.ptr(saved), C.c_size_t(sb if need_grad else 0),
                _lib.ptr(scratch), C.c_size_t(wb if fwd_scratch else 0), _lib.stream_ptr(dev), _lib.exec_ctx(dev)))
        # the layout of `saved` under the EQD_* switches in force NOW; the backward checks that they have not changed
        ctx.saved_layout = int(lib.eqd_model_saved_layout(C.byref(desc), C.byref(gs))) if need_grad else None
        packed._last_saved = (saved, sb, drop) if need_grad else None     # for IEGMN.layer_state (tests); freed with the batch
        ctx.drop = drop                 # the masks of THIS forward: the backward applies the same ones
        ctx.packed, ctx.desc, ctx.table_idx, ctx.saved, ctx.sb, ctx.wb = packed, desc, table_idx, saved, sb, wb
        ctx.tensors, ctx.ptrs = tensors, ptrs
        # (bf16 storage mode: the forward's transients lived in `scratch`; they are dead now, so the buffer is NOT kept for
        #  the backward - held on ctx across the loss, and across every other live forward, it gave back much of what the
        #  bf16 saved state saves; the caching allocator hands the same block to the backward's _workspace(wb))
        ctx.scratch = None
        ctx.flat_state = flat_state
        ctx.x0 = packed.x0      # keep the coordinates this forward used alive (the saved state points at them)
        ctx.mark_non_differentiable(status)
        return lig, Yl, Yr, T, b, status

    @staticmethod
    def backward(ctx, d_lig, d_Yl, d_Yr, d_T, d_b, _d_status):
        if ctx.saved is None:
            raise _lib.EquidockHipError("backward called on a forward that ran without saving state")
        lib = _lib.load_library()
        packed, desc = ctx.packed, ctx.desc
        dev = packed.x0.device
        if packed.x0 is not ctx.x0:      # a later forward re-read the coordinates: use this forward's, and make the
            packed.x0 = ctx.x0           # next forward re-read them again
            packed._x0_key = None
        gs = packed.c_struct()
        if int(lib.eqd_model_saved_layout(C.byref(desc), C.byref(gs))) != ctx.saved_layout:
            raise _lib.EquidockHipError("the EQD_* switches that decide the layout of the saved state changed between this "
                                        "forward and its backward (eqd_tunables_reload): run the forward again")
        tensors = ctx.tensors
        ptrs = ctx.ptrs
        if ctx.flat_state is not None:      # accumulate straight into the model's persistent flat buffer
            flat, offs, goffs = ctx.flat_state
        else:
            offs, total = flat_layout(tensors)
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            goffs = (C.c_int64 * len(ctx.table_idx))(*[offs[i] for i in ctx.table_idx])
        scratch = ctx.scratch if ctx.scratch is not None else _workspace(ctx.wb, dev)

        def prep(t):
            return None if t is None else _lib.require_device(t.to(torch.float32).contiguous(), 'output gradient')
        d_lig, d_Yl, d_Yr, d_T, d_b = (prep(t) for t in (d_lig, d_Yl, d_Yr, d_T, d_b))
        if _lib.profiling:
            lib.eqd_profile_mark(b'(between forward and backward: the caller\'s loss + its autograd)')
        dstruct = None if ctx.drop is None else ctx.drop.c_struct()
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_model_backward(
                C.byref(desc), C.byref(gs), ptrs, None if dstruct is None else C.byref(dstruct),
                _lib.ptr(d_lig), _lib.ptr(d_Yl), _lib.ptr(d_Yr), _lib.ptr(d_T),
                _lib.ptr(d_b), None, None, _lib.ptr(flat), goffs, _lib.ptr(ctx.saved), C.c_size_t(ctx.sb), _lib.ptr(scratch),
                C.c_size_t(ctx.wb), _lib.stream_ptr(dev), _lib.exec_ctx(dev)))
        if ctx.flat_state is not None:
            return (None,) * 9
        grads = tuple(flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, tensors))
        return (None, None, None, None, None, None, None) + grads


class IEGMN(nn.Module):

    def __init__(self, args, n_lays, fine_tune, log=None):
        super().__init__()
        self.fine_tune = fine_tune
        self.debug = args['debug']
        self.log = log
        self.device = args['device']
        self.graph_nodes = args['graph_nodes']
        self.rot_model = args['rot_model']
        self.noise_decay_rate = args['noise_decay_rate']
        self.noise_initial = args['noise_initial']
        self.use_edge_features_in_gmn = args['use_edge_features_in_gmn']
        self.use_mean_node_features = args['use_mean_node_features']
        self.n_lays = n_lays
        self.args = dict(args)

        self.residue_emb_layer = nn.Embedding(num_embeddings=21, embedding_dim=args['residue_emb_dim'])
        assert self.graph_nodes == 'residues'
        input_node_feats_dim = args['residue_emb_dim']
        if self.use_mean_node_features:
            input_node_feats_dim += 5
        self.iegmn_layers = nn.ModuleList()
        self.iegmn_layers.append(IEGMN_Layer(orig_h_feats_dim=input_node_feats_dim, h_feats_dim=input_node_feats_dim,
                                             out_feats_dim=args['iegmn_lay_hid_dim'], fine_tune=fine_tune, args=args,
                                             log=log))
        if args['shared_layers']:
            interm_lay = IEGMN_Layer(orig_h_feats_dim=input_node_feats_dim, h_feats_dim=args['iegmn_lay_hid_dim'],
                                     out_feats_dim=args['iegmn_lay_hid_dim'], args=args, fine_tune=fine_tune, log=log)
            for _ in range(1, n_lays):
                self.iegmn_layers.append(interm_lay)
        else:
            for _ in range(1, n_lays):
                self.iegmn_layers.append(IEGMN_Layer(orig_h_feats_dim=input_node_feats_dim,
                                                     h_feats_dim=args['iegmn_lay_hid_dim'],
                                                     out_feats_dim=args['iegmn_lay_hid_dim'], args=args,
                                                     fine_tune=fine_tune, log=log))
        assert args['rot_model'] == 'kb_att'
        self.num_att_heads = args['num_att_heads']
        self.out_feats_dim = args['iegmn_lay_hid_dim']
        self.att_mlp_key_ROT = nn.Sequential(
            nn.Linear(self.out_feats_dim, self.num_att_heads * self.out_feats_dim, bias=False))
        self.att_mlp_query_ROT = nn.Sequential(
            nn.Linear(self.out_feats_dim, self.num_att_heads * self.out_feats_dim, bias=False))
        self.mlp_h_mean_ROT = nn.Sequential(
            nn.Linear(self.out_feats_dim, self.out_feats_dim), nn.Dropout(args['dropout']),
            get_non_lin(args['nonlin'], args['leakyrelu_neg_slope']))
        self._flat = None               # (flat grad buffer, offsets, ids) once enable_flat_grads() is called
        self._table_cache = None
        self.svd_seed = 0
        self.svd_draws = None           # optional [B,10,3] tensor of guard perturbations (tests)
        self.last_svd_status = None     # int32 [B] device tensor: guard perturbations per pair (11 = unstable)

    # ---- C-ABI plumbing -------------------------------------------------------------------------
    def _desc(self):
        a = self.args
        d = _lib.EqdModelDesc()
        d.n_layers = self.n_lays
        d.d_emb = a['residue_emb_dim']
        d.d_hid = a['iegmn_lay_hid_dim']
        d.use_mean_node_features = int(bool(a['use_mean_node_features']))
        d.edge_feats = a['input_edge_feats_dim']
        d.n_heads = a['num_att_heads']
        d.cross_msgs = int(bool(a['cross_msgs']))
        d.use_dist_in_layers = int(bool(a['use_dist_in_layers']))
        d.use_edge_features = int(bool(a['use_edge_features_in_gmn']))
        d.skip_weight_h = a['skip_weight_h']
        d.x_connection_init = a['x_connection_init']
        d.lrelu_slope = a['leakyrelu_neg_slope']
        d.ln_eps = 1e-5
        d.svd_seed = int(self.svd_seed)
        # not a reference option: 'hip_storage_dtype' = 'bf16' runs EVERY GEMM of the IEGMN layers on the bf16 MFMA (edge
        # messages, node-level Linears and their weight gradients, all attention contractions: inputs rounded to bf16,
        # fp32 accumulate); coordinates, RBFs, LayerNorm / softmax statistics, the keypoint head and Kabsch stay fp32
        sd = a.get('hip_storage_dtype', 'fp32')
        if sd not in ('fp32', 'bf16'):
            raise NotImplementedError(f"hip_storage_dtype={sd!r}: only 'fp32' and 'bf16' exist")
        d.storage_bf16 = int(sd == 'bf16')
        return d

    def _param_table(self):
        """(unique parameters, table -> unique index).  Cached: Parameter objects persist across
        .to()/load_state_dict(); their data pointers are re-read at every call."""
        cached = self._table_cache
        if cached is not None and cached[0][0].device == cached[2] and cached[0][0].dtype == torch.float32:
            return cached[0], cached[1]
        uniq, table_idx = self._build_param_table()
        for t in uniq:
            _lib.require_device(t, 'parameter')
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.EquidockHipError("parameters must be contiguous fp32 tensors")
        self._table_cache = (uniq, table_idx, uniq[0].device)
        return uniq, table_idx

    def _build_param_table(self):
        table = []
        for lay in self.iegmn_layers:
            table.extend(lay.param_table())
        table.extend([self.residue_emb_layer.weight, self.att_mlp_key_ROT[0].weight,
                      self.att_mlp_query_ROT[0].weight, self.mlp_h_mean_ROT[0].weight, self.mlp_h_mean_ROT[0].bias])
        uniq, index, table_idx = [], {}, []
        for t in table:
            k = id(t)
            if k not in index:
                index[k] = len(uniq)
                uniq.append(t)
            table_idx.append(index[k])
        return uniq, table_idx

    # ---- flat gradient buffer (one buffer, one collective: SURVEY.md section 8e) -------------------
    def enable_flat_grads(self):
        """Make every parameter's .grad a view into ONE persistent fp32 buffer that the backward C call
        accumulates into directly (no per-parameter autograd work, one all-reduce for data parallel).
        zero_flat_grads() is one fill; optimizer.zero_grad() (either set_to_none) also works: dropped views are zeroed
        and restored at the next forward (_rebind_flat_views)."""
        uniq, _ = self._param_table()
        offs, total = flat_layout(uniq)
        flat = torch.zeros(total, dtype=torch.float32, device=uniq[0].device)
        self._flat_views = [flat[o:o + p.numel()].view(p.shape) for p, o in zip(uniq, offs)]
        for p, v in zip(uniq, self._flat_views):
            # frozen parameters keep .grad = None: their slice of the buffer still receives the C call's sums (the
            # kernels write every weight gradient), it is simply never shown to an optimizer
            p.grad = v if p.requires_grad else None
        _, table_idx = self._param_table()
        goffs = (C.c_int64 * len(table_idx))(*[offs[i] for i in table_idx])
        anchor = torch.zeros((), dtype=torch.float32, device=uniq[0].device, requires_grad=True)
        self._flat = (flat, offs, [id(p) for p in uniq], goffs, anchor)
        return flat

    def zero_flat_grads(self):
        self._flat[0].zero_()

    def _rebind_flat_views(self, uniq):
        """Keep every trainable parameter's .grad a view into the flat buffer.  The reference's training loop calls
        optimizer.zero_grad() (src/train.py:88), whose default set_to_none=True drops the views: a dropped view means
        "this gradient is zero now", so the slice is zeroed and the view restored (otherwise the backward would keep
        accumulating into a buffer nothing zeroes while the optimizer skips every parameter).  A .grad that was
        replaced by some other tensor is an error - the C call cannot accumulate into it."""
        flat, offs, views = self._flat[0], self._flat[1], self._flat_views
        dropped = []
        for i, p in enumerate(uniq):
            gr = p.grad
            if gr is views[i]:              # identity: the view object itself is still attached (the common case)
                continue
            if gr is None:
                if p.requires_grad:
                    dropped.append(i)
            elif gr.data_ptr() != flat.data_ptr() + 4 * offs[i]:
                raise _lib.EquidockHipError(
                    "a parameter's .grad no longer aliases the flat gradient buffer (it was replaced, not zeroed): "
                    "call enable_flat_grads() again, or use zero_flat_grads() / optimizer.zero_grad()")
        if not dropped:
            return
        whole = len(dropped) == sum(1 for p in uniq if p.requires_grad)
        if whole:
            flat.zero_()                    # zero_grad() dropped all of them: one fill
        for i in dropped:
            if not whole:
                views[i].zero_()
            uniq[i].grad = views[i]

    @property
    def grad_flat(self):
        return None if self._flat is None else self._flat[0]

    def run(self, batch_hetero_graph):
        """Returns the raw batched outputs (lig [n_lig,3], Yl, Yr [B,K,3], T [B,3,3], b [B,3])."""
        batch_hetero_graph = adapt_graph(batch_hetero_graph)
        if not self.uses_hip_path():
            from . import torch_path
            T, b, Yl, Yr, lig = torch_path.iegmn_forward(self, batch_hetero_graph)
            self.last_svd_status = None
            return None, lig, Yl, Yr, T, b      # no packed kernel layout on this path (and no pack() cost or constraints)
        packed = batch_hetero_graph.pack()
        uniq, table_idx = self._param_table()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in uniq)
        flat_state = None
        if self._flat is not None and need_grad:
            if self._flat[2] != [id(p) for p in uniq] or self._flat[0].device != packed.x0.device:
                raise _lib.EquidockHipError("parameters changed since enable_flat_grads(); call it again")
            self._rebind_flat_views(uniq)
            flat_state = (self._flat[0], self._flat[1], self._flat[3])
        # nn.Dropout is active while training with args['dropout'] > 0 (half of the published family's hyper-parameter
        # draws, src/utils/args.py:240): the masks come from torch's generator in the reference's order, the kernels
        # apply them (no torch-operator detour any more)
        drop = None
        if self.training and self.args['dropout'] > 0:
            drop = DropoutMasks.draw(self, batch_hetero_graph, packed)
        if flat_state is not None:
            lig, Yl, Yr, T, b, status = _IEGMNFunction.apply(packed, self._desc(), table_idx, self.svd_draws,
                                                             need_grad, flat_state, drop, uniq, self._flat[4])
        else:
            lig, Yl, Yr, T, b, status = _IEGMNFunction.apply(packed, self._desc(), table_idx, self.svd_draws,
                                                             need_grad, None, drop, *uniq)
        self.last_svd_status = status
        return packed, lig, Yl, Yr, T, b

    def layer_state(self, batch_hetero_graph, layer):
        """(h [n_nodes, width], x [n_nodes, 3]) after `layer` layers of the LAST forward of this batch that kept its state
        (a forward with gradients enabled): ligand nodes first, then receptor nodes - what the reference keeps as
        'hv_iegmn_out' / 'x_iegmn_out' for layer == n_lays (rigid_docking_model.py:507-510).  Copies; a test / debug aid."""
        packed = batch_hetero_graph.pack()
        last = getattr(packed, '_last_saved', None)
        if last is None:
            raise _lib.EquidockHipError("no saved forward state for this batch (run a forward with gradients enabled)")
        saved, sb = last[0], last[1]
        lib = _lib.load_library()
        desc, gs = self._desc(), packed.c_struct()
        hp, xp, w = C.c_void_p(), C.c_void_p(), C.c_int(0)
        _lib.check(lib.eqd_model_layer_state(C.byref(desc), C.byref(gs), _lib.ptr(saved), C.c_size_t(sb), int(layer),
                                             C.byref(hp), C.byref(w), C.byref(xp)))
        n = packed.n_nodes
        base = saved.data_ptr()
        f = saved.view(torch.float32) if saved.numel() % 4 == 0 else saved[:saved.numel() // 4 * 4].view(torch.float32)
        h = f[(hp.value - base) // 4:(hp.value - base) // 4 + n * w.value].view(n, w.value).clone()
        if layer == 0:
            x = packed.x0.clone()
        else:
            x = f[(xp.value - base) // 4:(xp.value - base) // 4 + n * 3].view(n, 3).clone()
        return h, x

    def stack_backward(self, batch_hetero_graph, d_h_last, d_x_last):
        """Backward of the IEGMN layer stack alone, from a given gradient w.r.t. the state after the last layer (d_h_last
        [n_nodes, 64], d_x_last [n_nodes, 3]; ligand nodes first): {parameter name: gradient} of
        sum(h_L * d_h_last) + sum(x_L * d_x_last) for the LAST forward of this batch that kept its state.  The keypoint /
        Kabsch head receives zero output gradients and contributes nothing.  A test aid (eqd_model_backward's d_h_last /
        d_x_last): in bf16 mode the head amplifies rounding flips erratically, the stack does not."""
        packed = batch_hetero_graph.pack()
        last = getattr(packed, '_last_saved', None)
        if last is None:
            raise _lib.EquidockHipError("no saved forward state for this batch (run a forward with gradients enabled)")
        saved, sb, drop = last
        lib = _lib.load_library()
        desc, gs = self._desc(), packed.c_struct()
        uniq, table_idx = self._param_table()
        dev = packed.x0.device
        ptrs = (C.c_void_p * len(table_idx))(*[uniq[i].data_ptr() for i in table_idx])
        offs, total = flat_layout(uniq)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        goffs = (C.c_int64 * len(table_idx))(*[offs[i] for i in table_idx])
        with _lib.device_guard(dev):
            wb = lib.eqd_model_scratch_bytes(C.byref(desc), C.byref(gs))
        scratch = torch.empty(wb, dtype=torch.uint8, device=dev)
        dh = _lib.require_device(d_h_last.to(torch.float32).contiguous(), 'd_h_last')
        dx = _lib.require_device(d_x_last.to(torch.float32).contiguous(), 'd_x_last')
        if tuple(dh.shape) != (packed.n_nodes, 64) or tuple(dx.shape) != (packed.n_nodes, 3):
            raise _lib.EquidockHipError("stack_backward: d_h_last must be [n_nodes, 64] and d_x_last [n_nodes, 3]")
        dstruct = None if drop is None else drop.c_struct()
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_model_backward(
                C.byref(desc), C.byref(gs), ptrs, None if dstruct is None else C.byref(dstruct), None, None, None, None,
                None, _lib.ptr(dh), _lib.ptr(dx), _lib.ptr(flat), goffs, _lib.ptr(saved), C.c_size_t(sb),
                _lib.ptr(scratch), C.c_size_t(wb), _lib.stream_ptr(dev), _lib.exec_ctx(dev)))
        names = {id(p): k for k, p in self.named_parameters()}
        return {names[id(p)]: flat[o:o + p.numel()].view(p.shape).clone() for p, o in zip(uniq, offs)}

    def head_backward(self, batch_hetero_graph, d_lig=None, d_Yl=None, d_Yr=None, d_T=None, d_b=None):
        """Backward of the keypoint / Kabsch head alone (eqd_model_head_backward) for the LAST forward of this batch that
        kept its state, from gradients w.r.t. the batched outputs (lig [n_lig, 3], Yl, Yr [B, K, 3], T [B, 3, 3], b [B, 3];
        None = zero): returns (d_h_L [n_nodes, 64], d_x_L [n_nodes, 3], {head parameter name: gradient}).  A test aid: with
        stack_backward it splits the whole-model gradient at the last layer's state, so that the head - fp32 in every mode -
        can be compared plainly in bf16 mode too."""
        packed = batch_hetero_graph.pack()
        last = getattr(packed, '_last_saved', None)
        if last is None:
            raise _lib.EquidockHipError("no saved forward state for this batch (run a forward with gradients enabled)")
        saved, sb, drop = last
        lib = _lib.load_library()
        desc, gs = self._desc(), packed.c_struct()
        uniq, table_idx = self._param_table()
        dev = packed.x0.device
        ptrs = (C.c_void_p * len(table_idx))(*[uniq[i].data_ptr() for i in table_idx])
        offs, total = flat_layout(uniq)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        goffs = (C.c_int64 * len(table_idx))(*[offs[i] for i in table_idx])
        with _lib.device_guard(dev):
            wb = lib.eqd_model_scratch_bytes(C.byref(desc), C.byref(gs))
        scratch = torch.empty(wb, dtype=torch.uint8, device=dev)

        def prep(t, what):
            return None if t is None else _lib.require_device(t.to(torch.float32).contiguous(), what)
        gl, gyl, gyr, gt, gb = (prep(t, w) for t, w in ((d_lig, 'd_lig'), (d_Yl, 'd_Yl'), (d_Yr, 'd_Yr'), (d_T, 'd_T'), (d_b, 'd_b')))
        n = packed.n_nodes
        d_h = torch.empty(n, 64, dtype=torch.float32, device=dev)
        d_x = torch.empty(n, 3, dtype=torch.float32, device=dev)
        dstruct = None if drop is None else drop.c_struct()

        def p(t):
            return None if t is None else _lib.ptr(t)
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_model_head_backward(
                C.byref(desc), C.byref(gs), ptrs, None if dstruct is None else C.byref(dstruct), p(gl), p(gyl), p(gyr), p(gt),
                p(gb), _lib.ptr(flat), goffs, _lib.ptr(saved), C.c_size_t(sb), _lib.ptr(scratch), C.c_size_t(wb),
                _lib.ptr(d_h), _lib.ptr(d_x), _lib.stream_ptr(dev)))
        names = {id(q): k for k, q in self.named_parameters()}
        grads = {names[id(q)]: flat[o:o + q.numel()].view(q.shape).clone() for q, o in zip(uniq, offs)
                 if 'iegmn_layers' not in names[id(q)] and 'residue_emb_layer' not in names[id(q)]}
        return d_h, d_x, grads

    def lrelu_signs(self, batch_hetero_graph):
        """The LeakyReLU branch decisions (uint8, 1 = pre-activation > 0) of the LAST forward of this batch that kept its
        state, in the library's node / edge order (ligand nodes then receptor nodes; PackedGraph edge order): a list with
        one dict per layer - 'edge_mlp', 'coors_mlp' [n_edges, 64], 'node_mlp', 'att_mlp_Q', 'att_mlp_K' [n_nodes, d_in] -
        followed by {'mlp_h_mean_ROT': [n_nodes, 64]}.  These are exactly the masks the backward applies
        (eqd_model_lrelu_signs); a test / debug aid: tests/parity_common.py hands them to the CPU oracle so that its
        gradient can be compared plainly even when a pre-activation lies within fp32 rounding of 0."""
        packed = batch_hetero_graph.pack()
        last = getattr(packed, '_last_saved', None)
        if last is None:
            raise _lib.EquidockHipError("no saved forward state for this batch (run a forward with gradients enabled)")
        saved, sb, drop = last
        dstruct = None if drop is None else drop.c_struct()
        dref = None if dstruct is None else C.byref(dstruct)
        lib = _lib.load_library()
        desc, gs = self._desc(), packed.c_struct()
        uniq, table_idx = self._param_table()
        ptrs = (C.c_void_p * len(table_idx))(*[uniq[i].data_ptr() for i in table_idx])
        dev = packed.x0.device
        N, E = packed.n_nodes, packed.n_edges
        d0 = self.args['residue_emb_dim'] + (5 if self.use_mean_node_features else 0)
        out = []

        def u8(*shape):
            return torch.zeros(*shape, dtype=torch.uint8, device=dev)
        with _lib.device_guard(dev):
            for l in range(self.n_lays):
                d = d0 if l == 0 else self.args['iegmn_lay_hid_dim']
                t = dict(edge_mlp=u8(E, 64), coors_mlp=u8(E, 64), node_mlp=u8(N, d))
                if self.args['cross_msgs']:
                    t.update(att_mlp_Q=u8(N, d), att_mlp_K=u8(N, d))
                _lib.check(lib.eqd_model_lrelu_signs(
                    C.byref(desc), C.byref(gs), ptrs, dref, _lib.ptr(saved), C.c_size_t(sb), l, _lib.ptr(t['edge_mlp']),
                    _lib.ptr(t['coors_mlp']), _lib.ptr(t['node_mlp']), _lib.ptr(t.get('att_mlp_Q')),
                    _lib.ptr(t.get('att_mlp_K')), _lib.stream_ptr(dev)))
                out.append(t)
            t = dict(mlp_h_mean_ROT=u8(N, 64))
            _lib.check(lib.eqd_model_lrelu_signs(
                C.byref(desc), C.byref(gs), ptrs, dref, _lib.ptr(saved), C.c_size_t(sb), self.n_lays, None, None,
                _lib.ptr(t['mlp_h_mean_ROT']), None, None, _lib.stream_ptr(dev)))
            out.append(t)
        return out

    def uses_hip_path(self):
        """The published family runs in the HIP library - in eval AND in training mode, with or without dropout; the
        reference's other options run through torch operators on the same device (hip_path_supported).  Decided by the
        configuration alone (`_force_torch_path` is a test aid: the torch-operator restatement of the same configuration)."""
        return hip_path_supported(self.args, self.fine_tune) and not getattr(self, '_force_torch_path', False)

    def forward(self, batch_hetero_graph, epoch):
        """[T_align list, b_align list, Y_ligand list, Y_receptor list] like the reference (:602)."""
        _, lig, Yl, Yr, T, b = self.run(batch_hetero_graph)
        B = Yl.shape[0]
        return [[T[i] for i in range(B)], [b[i].view(1, 3) for i in range(B)],
                [Yl[i] for i in range(B)], [Yr[i] for i in range(B)]]

    def __repr__(self):
        return f"IEGMN (HIP) n_lays={self.n_lays}"


class Rigid_Body_Docking_Net(nn.Module):

    def __init__(self, args, log=None):
        super().__init__()
        self.debug = args['debug']
        self.log = log
        self.device = args['device']
        self.iegmn_original = IEGMN(args, n_lays=args['iegmn_n_lays'], fine_tune=False, log=log)
        if args['fine_tune']:      # rigid_docking_model.py:622-627: a second, 2-layer stage on the moved ligand
            self.iegmn_fine_tune = IEGMN(args, n_lays=2, fine_tune=True, log=log)
            self.list_iegmns = [('original', self.iegmn_original), ('finetune', self.iegmn_fine_tune)]
        else:
            self.list_iegmns = [('finetune', self.iegmn_original)]

    def forward_batched(self, batch_hetero_graph):
        """Batched tensors instead of per-pair lists (no Python loop over pairs): lig [n_lig, 3], Yl, Yr [B, K, 3],
        T [B, 3, 3], b [B, 3]."""
        batch_hetero_graph = adapt_graph(batch_hetero_graph)
        out = None
        for stage, iegmn in self.list_iegmns:      # rigid_docking_model.py:646-682
            _, lig, Yl, Yr, T, b = iegmn.run(batch_hetero_graph)
            out = (lig, Yl, Yr, T, b)
            if stage == 'original':                # the fine-tune stage starts from the moved ligand (:667-669, 681-682)
                batch_hetero_graph = batch_hetero_graph.with_ligand_coords(lig)
        return out

    def forward(self, batch_hetero_graph, epoch):
        lig, Yl, Yr, T, b = self.forward_batched(batch_hetero_graph)
        B = Yl.shape[0]
        counts = [int(c) for c in (batch_hetero_graph.batch_num_nodes('ligand'))]
        ligs = list(torch.split(lig, counts, dim=0))
        return ligs, [Yl[i] for i in range(B)], [Yr[i] for i in range(B)], [T[i] for i in range(B)], \
            [b[i].view(1, 3) for i in range(B)]

    def __repr__(self):
        return "Rigid_Body_Docking_Net (HIP)"


def create_model(args, log=None):
    """src/utils/train_utils.py:111-114."""
    return Rigid_Body_Docking_Net(args=args, log=log)
