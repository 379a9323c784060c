"""Loss terms that follow the IEGMN forward in every training step of the reference (src/train.py:112-150), computed
for all pairs of a batch by libequidock_hip.so (eqd_pair_losses_fwd / _bwd, include/equidock_hip.h) instead of the
reference's Python loop with an (n_l x n_r) torch matrix per pair and term.

    mse, inter = pair_losses(batch, lig_pred, lig_target, rec, sigma, surface_ct)      # [B], [B]
    loss = mse.mean() + args['intersection_loss_weight'] * inter.mean()                # src/train.py:143-150

`compute_body_intersection_loss` keeps the reference's name and signature for one pair.  The pocket OT term
(compute_ot_emd -> ot.emd, src/utils/ot_utils.py:22-29) stays where the reference has it (host, POT): it is not part
of this library.  No CPU fallback: tensors must be on the GPU.
"""
import ctypes as C

import torch

from . import _lib
from .graph import PairGraph


class _SegmentsOnly:
    """Minimal stand-in for a PackedGraph when only the pair segmentation matters (the loss kernels read n_pairs,
    n_lig and seg_off of EqdGraph, nothing else): ONE pair of n_l ligand and n_r receptor nodes."""

    def __init__(self, n_l, n_r, device):
        self.n_pairs, self.n_lig, self.n_rec = 1, int(n_l), int(n_r)
        self.seg_off = torch.tensor([0, n_l, n_l + n_r], dtype=torch.int32, device=device)
        self.x0 = self.seg_off          # (device carrier)
        self._gs = None

    def c_struct(self):
        if self._gs is None:
            g = _lib.EqdGraph()
            g.n_pairs, g.n_lig, g.n_rec, g.n_nodes = 1, self.n_lig, self.n_rec, self.n_lig + self.n_rec
            _lib.require_device(self.seg_off, 'seg_off')
            g.seg_off = self.seg_off.data_ptr()
            self._gs = g
        return self._gs


class _PairLosses(torch.autograd.Function):

    @staticmethod
    def forward(ctx, packed, lig_pred, lig_target, rec, sigma, surface_ct):
        lib = _lib.load_library()
        dev = packed.x0.device
        f32 = dict(dtype=torch.float32, device=dev)

        def prep(t, n, what):
            t = _lib.require_device(t.detach().to(torch.float32).contiguous(), what)
            if tuple(t.shape) != (n, 3):
                raise _lib.EquidockHipError(f"{what}: expected shape ({n}, 3), got {tuple(t.shape)}")
            return t
        a = prep(lig_pred, packed.n_lig, 'lig_pred')
        t = prep(lig_target, packed.n_lig, 'lig_target')
        r = prep(rec, packed.n_rec, 'rec')
        B = packed.n_pairs
        mse, inter = torch.empty(B, **f32), torch.empty(B, **f32)
        s_lig, s_rec = torch.empty(packed.n_lig, **f32), torch.empty(packed.n_rec, **f32)
        gs = packed.c_struct()
        _lib.check(lib.eqd_pair_losses_fwd(C.byref(gs), _lib.ptr(a), _lib.ptr(t), _lib.ptr(r), C.c_float(sigma),
                                           C.c_float(surface_ct), _lib.ptr(mse), _lib.ptr(inter), _lib.ptr(s_lig),
                                           _lib.ptr(s_rec), _lib.stream_ptr(dev)))
        ctx.packed, ctx.sigma, ctx.ct = packed, float(sigma), float(surface_ct)
        ctx.save_for_backward(a, t, r, s_lig, s_rec)
        ctx.set_materialize_grads(False)
        return mse, inter

    @staticmethod
    def backward(ctx, d_mse, d_inter):
        lib = _lib.load_library()
        a, t, r, s_lig, s_rec = ctx.saved_tensors
        packed = ctx.packed
        dev = a.device
        d_a = torch.empty_like(a)

        def prep(g):
            return None if g is None else g.to(torch.float32).contiguous()
        d_mse, d_inter = prep(d_mse), prep(d_inter)
        gs = packed.c_struct()
        _lib.check(lib.eqd_pair_losses_bwd(C.byref(gs), _lib.ptr(a), _lib.ptr(t), _lib.ptr(r), C.c_float(ctx.sigma),
                                           C.c_float(ctx.ct), _lib.ptr(s_lig), _lib.ptr(s_rec), _lib.ptr(d_mse),
                                           _lib.ptr(d_inter), _lib.ptr(d_a), _lib.stream_ptr(dev)))
        return None, d_a, None, None, None, None


def pair_losses(batch, lig_pred, lig_target, rec, sigma, surface_ct):
    """Per-pair (MSE of the predicted ligand coordinates, body-intersection loss) for a PairGraph batch.
    lig_pred / lig_target: [n_lig, 3] in the batch's ligand node order (e.g. the first output of
    Rigid_Body_Docking_Net.forward_batched); rec: [n_rec, 3] bound receptor coordinates.  Gradients flow to lig_pred."""
    if not isinstance(batch, PairGraph):
        raise TypeError("expected an equidock_public_amd.graph.PairGraph batch")
    return _PairLosses.apply(batch.pack(), lig_pred, lig_target, rec, float(sigma), float(surface_ct))


def compute_body_intersection_loss(model_ligand_coors_deform, bound_receptor_repres_nodes_loc_array, sigma, surface_ct):
    """src/train.py:46-49 for ONE pair (a one-pair batch through the same kernels)."""
    a, r = model_ligand_coors_deform, bound_receptor_repres_nodes_loc_array
    g = _SegmentsOnly(a.shape[0], r.shape[0], a.device)
    _, inter = _PairLosses.apply(g, a, a.detach(), r, float(sigma), float(surface_ct))
    return inter[0]


class ScalarLoss:
    """Fixed scalar loss of the measurement harness: sum over pairs of mean(lig'^2) + mean(Yl^2) +
    mean(Yr^2) on the batched model outputs, value AND gradients w.r.t. the outputs in ONE launch (eqd_scalar_loss).

        sl = ScalarLoss(packed, n_heads)
        loss, grads = sl(lig, Yl, Yr)                       # 0-d tensor, (d_lig, d_Yl, d_Yr)
        torch.autograd.backward([lig, Yl, Yr], grads)        # == loss.backward() of the torch expression

    Buffers are allocated once, so the call is capturable in a hipGraph."""

    def __init__(self, packed, n_heads):
        dev = packed.x0.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.packed, self.K = packed, int(n_heads)
        B = packed.n_pairs
        self.d_lig = torch.empty(packed.n_lig, 3, **f32)
        self.d_Yl = torch.empty(B, self.K, 3, **f32)
        self.d_Yr = torch.empty(B, self.K, 3, **f32)
        self.pair_loss = torch.empty(B, **f32)
        self.loss = torch.zeros((), **f32)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)

    def __call__(self, lig, Yl, Yr):
        lib = _lib.load_library()
        dev = self.d_lig.device
        for t, ref in ((lig, self.d_lig), (Yl, self.d_Yl), (Yr, self.d_Yr)):
            _lib.require_device(t, 'model output')
            if t.shape != ref.shape or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.EquidockHipError("ScalarLoss: outputs must be contiguous fp32 tensors of the batch's shapes")
        gs = self.packed.c_struct()
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_scalar_loss(C.byref(gs), self.K, _lib.ptr(lig.detach()), _lib.ptr(Yl.detach()),
                                           _lib.ptr(Yr.detach()), _lib.ptr(self.d_lig), _lib.ptr(self.d_Yl),
                                           _lib.ptr(self.d_Yr), _lib.ptr(self.pair_loss), _lib.ptr(self.loss),
                                           _lib.ptr(self.counter), _lib.stream_ptr(dev)))
        return self.loss, (self.d_lig, self.d_Yl, self.d_Yr)
