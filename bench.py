"""bench.py -- protein-pairs/sec (fwd+bwd) of one IEGMN stack on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload B|A|C|D|E|R] [--dtype f32|bf16]

A "step" = zero grads -> Rigid_Body_Docking_Net forward -> fixed scalar loss
-> backward, on one synthetic batch that is already resident in HBM (the slice
src/train.py:88-100,154 of the reference without losses/OT/clip/optimizer).  Default workload =
BASELINE.json configs[1]: 8 pairs x (200, 200) residues, k = 10, 8-layer IEGMN, hdim 64, 50
heads, fp32, on ONE GPU.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
every rank runs the same per-GPU batch (weak scaling) and the flat gradient buffer is
all-reduced once per step over RCCL.

The default workload is B for every N, so that the driver's N = 1, 2, 4, 8 runs form one weak-scaling curve;
`--workload D` is BASELINE.json configs[3] (64 x (300,300) per GPU, bf16); `--workload R` is SURVEY.md section 8d's
"realistic sizes" variant: 64 ragged pairs drawn from the DB5.5 size statistics (ligand 29..1500, receptor 40..2130
residues; synthetic.realistic_sizes), which also reports residue-pairs/s (sum n_lig * n_rec per second: the quadratic
attention term) and the size spread.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":      the dominant kernel (edge-message backward/forward), measured live with HIP events on the launch
                   stream: algorithmic FLOPs|bytes per launch / avg duration of batched standalone launches;
  "roofline_all":  every kernel family >= 3 % of the step, timed inside an eager step with the library's launch
                   profiler (eqd_profile_*): executed and as-written FLOP rates, algorithmic-bytes HBM rate, PMC MfmaUtil;
  "whole_step":    as-written FLOPs of the step / ms_per_step against the MFMA peak of the dtype;
  "cpu_baseline":  the oracle (oracle/iegmn_port.py, the reference's op sequence) on the host's physical cores, plus
                   its one-thread and one-pair-per-step variants (N=1 only; the only place `oracle` is imported);
  "secondary":     (default workload B only) the other BASELINE.json configs timed the same way with 3 + 10 steps:
                   C in bf16 (configs[2]), E (configs[4]) and the ragged workload R on one GPU; D (configs[3]: 64 x (300,300)
                   per GPU in bf16, RCCL all-reduce inside the step graph) when launched by torch.distributed.run.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (pairs per GPU, (n_lig, n_rec), layers, shared, skip_weight_h, default dtype, description)
    'A': (1, (200, 200), 5, True, 0.5, 'f32', 'A: DB5.5 single pair (200,200), k=10, 5-layer shared IEGMN hdim 64, 50 heads, fp32'),
    'B': (8, (200, 200), 8, False, 0.75, 'f32', 'B: DB5.5-sized batch of 8 pairs x (200,200) residues, k=10, 8-layer IEGMN hdim 64, 50 heads, fp32'),
    'C': (64, (300, 300), 8, False, 0.75, 'f32', 'C: DIPS-sized batch of 64 pairs x (300,300) residues, k=10, 8-layer IEGMN'),
    'D': (64, (300, 300), 8, False, 0.75, 'bf16', 'D: DIPS-sized batch of 64 pairs x (300,300) per GPU (512 pairs on 8 GPUs), k=10, 8-layer IEGMN, data parallel'),
    'E': (4, (2000, 2000), 8, False, 0.75, 'f32', 'E: stress, 4 pairs x (2000,2000) residues, k=10, 8-layer IEGMN, fp32'),
    'R': (64, None, 8, False, 0.75, 'f32', 'R: ragged batch of 64 pairs with DB5.5-distributed sizes (ligand 29..1500, receptor 40..2130 residues; SURVEY.md section 8d "realistic sizes"), k=10, 8-layer IEGMN, fp32'),
}
R_SIZE_SEED = 5055      # synthetic.realistic_sizes(64, R_SIZE_SEED + rank): the same ragged batch in every run
PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md; no sparsity)
PEAK_HBM_GBS = 8000.0      # HBM3E spec
EDGE_FWD_FLOP_PER_EDGE = 38272          # SURVEY.md section 8d, layers >= 1, model as written
EDGE_FWD_BYTES = lambda n, e: n * 540 + e * 112   # noqa: E731  SURVEY.md section 8d (fp32)


def step_flops_as_written(sizes, L, d0=69, d=64, K=50):
    """algorithmic FLOPs of ONE forward of the model as written (2 per MAC) for the given pair
    sizes; forward + backward = 3x.  k = 10 in-edges per node."""
    tot = 0.0
    for nl, nr in sizes:
        n, e = nl + nr, 10 * (nl + nr)
        tot += e * 39552 + n * 74796 + 552.0 * nl * nr                       # layer 0 (d_in = 69)
        tot += (L - 1) * (e * 38272 + n * 66176 + 512.0 * nl * nr)            # layers >= 1
        tot += n * 424492 + 0.82e6                                            # keypoint head
    return tot


def kernel_work_model(sizes, L, n_nodes, n_edges, cross=True, d0=69, dh=64, d_emb=64, K=50, fused_fwd=False, rowwave='k_rowres',
                      fused_gather=False, ds_handoff=False, wide_resident=False):
    """Per-STEP work of each kernel family, from the launch structure of eqd_model_forward / eqd_model_backward
    (csrc/eqd_driver.hip): `flops` = FLOPs the kernel executes on the MFMA pipes for its GEMMs (2 per MAC),
    `flops_written` = the model-as-written share where SURVEY.md section 8d defines one (edge kernels, attention),
    `bytes` = algorithmic HBM bytes (every operand read once, every result written once; fp32).  DESIGN.md section 6
    spells the formulas out."""
    N, E = float(n_nodes), float(n_edges)
    pp = sum(float(a) * b for a, b in sizes)              # sum over pairs of n_lig * n_rec
    W = {k: dict(flops=0.0, flops_written=0.0, bytes=0.0) for k in
         ('k_linear', 'k_rowchain', 'k_rowwave', 'k_rowres', 'k_attn_fwd', 'k_attn_bwd', 'k_edge_fwd', 'k_edge_bwd', 'k_node_gather',
          'k_atb', 'k_atb_reduce', 'k_edge_attn_fwd', 'k_attn_bwd_gather', 'k_keypoint', 'k_keypoint_bwd', 'k_keypoint_bwd_a', 'k_keypoint_bwd_b',
          'k_head_u', 'k_head_u_bwd', 'k_attn_bwd_kvds', 'k_attn_bwd_qds', 'k_kabsch_fwd', 'k_kabsch_bwd', 'k_embed_fwd',
          'k_embed_bwd', 'k_seg_mean', 'k_qmean_bwd', 'k_reduce_segments')}
    # k_edge_attn_fwd: the 64-wide layers' edge + attention forward in one launch (small batches); k_attn_bwd_gather: the
    # 64-wide layers' attention backward with the edge backward's node gather (+ the partial reductions) in the same launch
    for l in range(L):
        d = d0 if l == 0 else dh
        da = (d + 15) // 16 * 16
        # from 4 tiles per CU the node-level jobs of the 64-wide layers run on k_rowres (weights resident in LDS; `rowwave`
        # = the name seen in the profile, k_rowwave when forced), layer 0's 69-wide ones and small batches on k_linear /
        # k_rowchain (csrc/eqd_node_kernels.hip: rw_mode, rw_eligible)
        # (wide_resident: bf16 mode at large sizes - the 69-wide first layer's jobs run on k_rowres80, profiled as k_rowres)
        lin = rowwave if (rowwave and (d == 64 or wide_resident)) else 'k_linear'
        chain = rowwave if (rowwave and (d == 64 or wide_resident)) else 'k_rowchain'
        # forward: five node projections (P, Q 64 wide; q, k, v d wide)
        W[lin]['flops'] += N * 2 * d * (128 + 3 * d)
        W[lin]['bytes'] += N * 4 * (d + 128 + 3 * da)
        # forward row chain: node_mlp.0 (+LeakyReLU, LayerNorm) -> node_mlp.4 (+ skip)
        W[chain]['flops'] += N * (2 * (d0 + 2 * d + 64) * d + 2 * d * 64)
        W[chain]['bytes'] += N * 4 * (d + 64 + da + d0 + 2 * d + 64)
        # backward row chain: dh of the layer above (6 transposed GEMMs), d a1n, LN/LReLU backward, 3 input gradients
        if l < L - 1:
            W[chain]['flops'] += N * 2 * 64 * (64 * 3 + 64 * 3)
            W[chain]['bytes'] += N * 4 * (64 * 3 + 80 * 3 + 64 + 64)
        W[chain]['flops'] += N * (2 * 64 * d + 2 * d * (64 + d + d_emb))
        W[chain]['bytes'] += N * 4 * (64 + d + d + 64 + da + d0)
        # attention (model as written: 8 Nl Nr d forward, 2x that backward; executed backward: S and dP are recomputed by
        # both the dq and the dk/dv workgroups -> 28 Nl Nr d)
        fa = 'k_edge_attn_fwd' if (fused_fwd and d == 64) else 'k_attn_fwd'
        fe_ = 'k_edge_attn_fwd' if (fused_fwd and d == 64) else 'k_edge_fwd'
        if cross:
            W[fa]['flops'] += 8 * pp * da
            W[fa]['flops_written'] += 8 * pp * d
            W[fa]['bytes'] += N * 4 * (4 * da + 1)
            if ds_handoff and d == 64:
                # dS hand-off (large batches): the key / value pass executes S, dP, dV, dK (4 units of 4 pp d) and writes dS
                # (2 pp floats), the dq pass contracts it with K (1 unit): 20 pp d executed for 16 as written
                W['k_attn_bwd_kvds']['flops'] += 16 * pp * da
                W['k_attn_bwd_kvds']['flops_written'] += 12 * pp * d
                W['k_attn_bwd_kvds']['bytes'] += N * 4 * (7 * da + 2) + 8 * pp
                W['k_attn_bwd_qds']['flops'] += 4 * pp * da
                W['k_attn_bwd_qds']['flops_written'] += 4 * pp * d
                W['k_attn_bwd_qds']['bytes'] += N * 4 * (3 * da) + 8 * pp
            else:
                ab = 'k_attn_bwd_gather' if (fused_gather and d == 64) else 'k_attn_bwd'
                W[ab]['flops'] += 28 * pp * da
                W[ab]['flops_written'] += 16 * pp * d
                W[ab]['bytes'] += N * 4 * (8 * da + 2)
        # edge kernels (as written: 2 (2 d + 42) 64 + 2 64 64 + 2 64 64 + 2 64 per edge; executed: the P/Q split moves
        # 2 (2 d) 64 per edge to k_linear - the executed count comes from the PMC file when present)
        fe = 2 * (2 * d + 42) * 64 + 2 * 64 * 64 + 2 * 64 * 64 + 2 * 64
        fx = 2 * 42 * 64 + 2 * 64 * 64 + 2 * 64 * 64 + 2 * 64
        W[fe_]['flops_written'] += E * fe
        W[fe_]['flops'] += E * fx
        W[fe_]['bytes'] += N * 540 + E * 112
        W['k_edge_bwd']['flops_written'] += 2 * E * fe
        W['k_edge_bwd']['flops'] += E * (3 * fx - 2 * 42 * 64)      # recompute + data gradients + weight gradients
        W['k_edge_bwd']['bytes'] += 2 * (N * 540 + E * 112)
        gk = 'k_attn_bwd_kvds' if (ds_handoff and cross and d == 64) else \
            ('k_attn_bwd_gather' if (fused_gather and cross and d == 64) else 'k_node_gather')
        W[gk]['bytes'] += E * 272 + N * 540      # (the layer's node gather rides in that launch)
        # end-of-pass weight-gradient GEMMs of the node-level Linears
        W['k_atb']['flops'] += N * 2 * (dh * d + d * d + d * 64 + (d * d if cross else 0) + d * d0 + 2 * 64 * d
                                        + (3 * d * d if cross else 0))
        W['k_atb']['bytes'] += N * 4 * (dh + d + d + 64 + da + d0 + 128 + d + 3 * da)
    # head: mlp_h_mean_ROT forward, its data gradient (64 wide), the final dh job of layer 0 (69-wide sources)
    W[rowwave or 'k_linear']['flops'] += N * (2 * 64 * 64 * 2)
    W[rowwave or 'k_linear']['bytes'] += N * 4 * (64 * 4 + 64)
    W['k_linear']['flops'] += N * (2 * d_emb * (4 * d0 + 128))
    W['k_linear']['bytes'] += N * 4 * (3 * 80 + 128 + d0)
    W['k_atb']['flops'] += N * 2 * 64 * 64
    # second stage of the weight-gradient GEMMs: the per-part partial outputs summed (bytes = what the first stage wrote,
    # taken from the PMC passes only - the part count is a launch-shape choice, no algorithmic figure exists for it)
    # keypoint head (6.4 kFLOP / node; N (64 + K + 3) 4 bytes): scores of K heads per node, softmax
    # over the protein's nodes, keypoints; backward twice that
    W['k_keypoint']['flops'] += N * 2 * 64 * K
    W['k_keypoint']['bytes'] += N * 4 * (64 + K + 3)
    # (the matrix-product backward, one launch: du and dH products; reads h, scores, z, writes dH, dz - no dscores array)
    W['k_keypoint_bwd']['flops'] += 2 * N * 2 * 64 * K
    W['k_keypoint_bwd']['bytes'] += N * 4 * (64 + K + 3 + 64 + 3)
    W['k_keypoint_bwd_a']['flops'] += N * 2 * 64 * K
    W['k_keypoint_bwd_a']['bytes'] += N * 4 * (64 + 2 * K + 3)
    W['k_keypoint_bwd_b']['flops'] += N * 2 * 64 * K
    W['k_keypoint_bwd_b']['bytes'] += N * 4 * (64 + K + 64)
    nb2 = 2.0 * len(sizes)
    W['k_head_u']['flops'] += nb2 * K * 2 * 2 * 64 * 64
    W['k_head_u']['bytes'] += K * 2 * 64 * 64 * 4 + nb2 * K * 64 * 4 * 2
    W['k_head_u_bwd']['flops'] += nb2 * K * 2 * 4 * 64 * 64
    W['k_head_u_bwd']['bytes'] += K * 2 * 64 * 64 * 4 * 2 + nb2 * K * 64 * 4 * 3
    # the small per-pair / per-node kernels around the head (bytes only: they are launch-latency-bound at every size)
    nlig = float(sum(a for a, _ in sizes))
    W['k_kabsch_fwd']['bytes'] += nb2 * K * 3 * 4 + nlig * 3 * 4 * 2 + len(sizes) * (9 + 3 + 9 + 21 * 2) * 4
    W['k_kabsch_bwd']['bytes'] += nb2 * K * 3 * 4 * 2 + nlig * 3 * 4 * 2 + len(sizes) * (9 + 3 + 9 + 21 * 2) * 4
    W['k_embed_fwd']['bytes'] += N * 4 * (1 + 5 + d0)
    W['k_embed_bwd']['bytes'] += N * 4 * (1 + 2 * d_emb) + 21 * d_emb * 4
    W['k_seg_mean']['bytes'] += N * 4 * 64 + nb2 * 64 * 4
    W['k_qmean_bwd']['bytes'] += nb2 * K * 64 * 4 + N * 64 * 4
    # deferred partial sums (LayerNorm / bias / coordinate-MLP vectors, embedding table): what the row chains and the edge
    # backward wrote per workgroup that did not ride in a gather launch
    W['k_reduce_segments']['bytes'] += L * (N / 16.0) * 256 * 4
    # second stage of the weight-gradient GEMMs: ~10 units per layer, each split over ~1024 / 36 row parts (more from 16 k
    # rows: csrc/eqd_node_kernels.hip, atb_units) whose 64 x 64 partial outputs it sums - a launch-shape figure, nominal
    W['k_atb_reduce']['bytes'] += (10 * L + 1) * max(28.0, N / 1024.0) * 64 * 64 * 4
    return W


class _BatchedLoss(torch.autograd.Function):
    """sum over pairs of mean(lig'^2) + mean(Yl^2) + mean(Yr^2) on the batched outputs (lig_w[i] = 1 / (3 n_pair(i))).

    The same scalar as oracle.iegmn_port.scalar_loss, written as ONE weighted sum of squares over the concatenated
    outputs with the backward spelled out: differentiated by autograd from the obvious expression it is 27 elementwise /
    reduction launches (~100 us of GPU time at workload B, 6 % of the step, none of it the path being measured); this
    way it is 6."""

    @staticmethod
    def forward(ctx, lig, Yl, Yr, lig_w):
        c = 1.0 / (Yl.shape[1] * Yl.shape[2])
        key = (lig_w.data_ptr(), Yl.shape, Yr.shape)
        if _BatchedLoss._wcat is None or _BatchedLoss._wcat[0] != key:      # weights of the concatenation, built once
            w = torch.cat([lig_w.expand(-1, 3).reshape(-1), torch.full((Yl.numel() + Yr.numel(),), c, dtype=lig_w.dtype,
                                                                       device=lig_w.device)])
            _BatchedLoss._wcat = (key, w)
        w = _BatchedLoss._wcat[1]
        v = torch.cat([lig.reshape(-1), Yl.reshape(-1), Yr.reshape(-1)])
        vw = v * w
        ctx.save_for_backward(vw)
        ctx.shapes = (lig.shape, Yl.shape, Yr.shape)
        return (vw * v).sum()

    @staticmethod
    def backward(ctx, g):
        (vw,) = ctx.saved_tensors
        gv = vw * (g * 2.0)
        sl, syl, syr = ctx.shapes
        nl, nyl = sl.numel(), syl.numel()
        return gv[:nl].view(sl), gv[nl:nl + nyl].view(syl), gv[nl + nyl:].view(syr), None


_BatchedLoss._wcat = None


def batched_loss(lig, Yl, Yr, lig_w):
    return _BatchedLoss.apply(lig, Yl, Yr, lig_w)


def _profile_order(path):
    """Sort key that puts the NEWEST committed counter summary last: the file's own "collected" stamp when it has one
    (json key `_collected`, ISO date-time written by profiles/merge_pmc.py), else its round number and name (r02_z4 after
    r02_z: plain name order gets that wrong, '_' sorts after digits) - mtimes do not survive a git checkout."""
    import re
    try:
        stamp = json.load(open(path)).get('_collected', '')
    except Exception:
        stamp = ''
    m = re.match(r'r(\d+)_([a-z]+)(\d*)', os.path.basename(path))
    rnd, tag, num = (int(m.group(1)), m.group(2), int(m.group(3) or 0)) if m else (0, '', 0)
    return (stamp, rnd, tag, num, os.path.basename(path))


def time_kernel(fn, iters, stream_sync):
    """Average duration of ONE launch of fn, from HIP events on the launch stream around batches of 10 back-to-back
    launches.  A spin kernel in front of every batch keeps the GPU busy while the host enqueues the batch, so host launch
    overhead (ctypes, Python; 5-15 us per launch depending on the box) is not in the figure - one event pair per launch
    measured 37-41 us for a kernel that rocprofv3 times at 34 us."""
    for _ in range(3):
        fn()
    stream_sync()
    per = 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(1, iters // per))]
    for e0, e1 in ev:
        torch.cuda._sleep(2_000_000)        # ~1 ms of spinning
        e0.record()
        for _ in range(per):
            fn()
        e1.record()
    ev[-1][1].synchronize()
    ts = [e0.elapsed_time(e1) / per for e0, e1 in ev]
    return sum(ts) / len(ts) * 1e-3   # seconds per launch


def edge_kernel_rooflines(net, packed, dev, workload='B', bf16=False):
    """Launch the edge-message kernels standalone (layer-1 weights, the workload's graph) on torch's
    current stream and time them with HIP events on that stream."""
    from equidock_public_amd import _lib
    lib = _lib.load_library()
    gs = _lib.graph_struct(packed)
    lay = net.iegmn_original.iegmn_layers[1]
    N, E = packed.n_nodes, packed.n_edges
    f = dict(dtype=torch.float32, device=dev)
    P, Q = torch.randn(N, 64, **f), torch.randn(N, 64, **f)
    x = packed.x0.clone()
    aggr, xnew = torch.empty(N, 64, **f), torch.empty(N, 3, **f)
    ep = _lib.EqdEdgeParams()
    W1 = lay.edge_mlp[0].weight
    ep.W1, ep.ldw1, ep.d_in = W1.data_ptr(), W1.shape[1], 64
    ep.ln_g, ep.ln_b = lay.edge_mlp[3].weight.data_ptr(), lay.edge_mlp[3].bias.data_ptr()
    ep.W2, ep.b2 = lay.edge_mlp[4].weight.data_ptr(), lay.edge_mlp[4].bias.data_ptr()
    ep.Wc1, ep.bc1 = lay.coors_mlp[0].weight.data_ptr(), lay.coors_mlp[0].bias.data_ptr()
    ep.wc2, ep.bc2 = lay.coors_mlp[4].weight.data_ptr(), lay.coors_mlp[4].bias.data_ptr()
    ep.slope, ep.ln_eps, ep.eta, ep.use_dist, ep.use_he = 0.01, 1e-5, 0.0, 1, 1
    ep.bf16 = int(bf16)
    st = _lib.stream_ptr(dev)
    sync = lambda: torch.cuda.current_stream(dev).synchronize()  # noqa: E731

    def fwd():
        _lib.check(lib.eqd_edge_message_fwd(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q), _lib.ptr(x),
                                            _lib.ptr(aggr), _lib.ptr(xnew), st))
    t_fwd = time_kernel(fwd, 50, sync)

    # backward kernel alone (k_edge_bwd): the C entry point also runs the weight-gradient GEMMs and the
    # CSC gather, so time the entry point and report the kernel's share from the profile separately.
    wsb = lib.eqd_edge_message_bwd_workspace_bytes(C.byref(gs))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dag, dxn = torch.randn(N, 64, **f), torch.randn(N, 3, **f)
    dP, dQ, dx = torch.empty(N, 64, **f), torch.empty(N, 64, **f), torch.empty(N, 3, **f)
    g = {k: torch.zeros_like(v) for k, v in (('W1', W1), ('lng', lay.edge_mlp[3].weight), ('lnb', lay.edge_mlp[3].bias),
                                             ('W2', lay.edge_mlp[4].weight), ('b2', lay.edge_mlp[4].bias),
                                             ('Wc1', lay.coors_mlp[0].weight), ('bc1', lay.coors_mlp[0].bias),
                                             ('wc2', lay.coors_mlp[4].weight), ('bc2', lay.coors_mlp[4].bias))}
    eg = _lib.EqdEdgeGrads()
    eg.dW1, eg.ldw1 = g['W1'].data_ptr(), W1.shape[1]
    eg.dln_g, eg.dln_b, eg.dW2, eg.db2 = (g[k].data_ptr() for k in ('lng', 'lnb', 'W2', 'b2'))
    eg.dWc1, eg.dbc1, eg.dwc2, eg.dbc2 = (g[k].data_ptr() for k in ('Wc1', 'bc1', 'wc2', 'bc2'))

    def bwd_op():
        _lib.check(lib.eqd_edge_message_bwd(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q), _lib.ptr(x),
                                            _lib.ptr(dag), _lib.ptr(dxn), _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dx),
                                            C.byref(eg), _lib.ptr(ws), C.c_size_t(wsb), st))

    def bwd():
        _lib.check(lib.eqd_edge_message_bwd_kernel_only(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q),
                                                        _lib.ptr(x), _lib.ptr(dag), _lib.ptr(dxn), _lib.ptr(dQ),
                                                        _lib.ptr(dx), _lib.ptr(ws), C.c_size_t(wsb), st))
    t_bwd = time_kernel(bwd, 30, sync)
    t_bwd_op = time_kernel(bwd_op, 20, sync)
    flop_f = EDGE_FWD_FLOP_PER_EDGE * E
    byte_f = EDGE_FWD_BYTES(N, E)
    out = {}
    if bf16:
        byte_f = N * 540 + E * (27 * 2 + 4)       # he rows in bf16
    peak_tf = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
    for name, t, mult in (('k_edge_fwd', t_fwd, 1), ('k_edge_bwd', t_bwd, 2)):
        tf = flop_f * mult / t / 1e12
        gb = byte_f * mult / t / 1e9
        out[name] = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak_tf, "unit": "TFLOP/s",
                     "frac": round(tf / peak_tf, 4), "traffic": None,
                     "avg_launch_us": round(t * 1e6, 2), "algorithmic_flops_per_launch": flop_f * mult,
                     "algorithmic_bytes_per_launch": byte_f * mult,
                     "hbm_algorithmic_GBps": round(gb, 1), "hbm_frac_algorithmic": round(gb / PEAK_HBM_GBS, 4)}
        if bf16 and gb / PEAK_HBM_GBS > tf / peak_tf:
            # with the GEMMs at the bf16 MFMA rate the HBM side is the closer roof of the two
            out[name].update({"bound": "hbm", "achieved": round(gb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(gb / PEAK_HBM_GBS, 4), "mfma_TFLOPs": round(tf, 3),
                              "mfma_frac_bf16_peak": round(tf / peak_tf, 4)})
    out['k_edge_bwd']['whole_bwd_op_us'] = round(t_bwd_op * 1e6, 2)   # + weight-grad GEMMs, reductions, CSC gather
    # HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json; counters cannot be read
    # from inside the process): (2 * FETCH_SIZE + WRITE_SIZE) KB, see the file's comment for the correction
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*traffic.json')), key=_profile_order)      # newest last
        # (bf16 runs have their own counter passes: key '<workload>_bf16')
        wkey = workload + ('_bf16' if bf16 else '')
        full = {} if not files else json.load(open(files[-1]))
        tr = full.get(wkey, {})
        stale = staleness(full.get('_workloads', {}).get(wkey, {}).get('csrc_digest'))
        if 'k_edge_fwd' not in tr and 'k_edge_attn_fwd' in tr:
            # DB5.5-sized batches run the edge forward inside the fused edge + attention launch: its counters are the only
            # measured traffic of the message forward there (they include the attention half's q / k / v / out rows)
            tr = dict(tr, k_edge_fwd=dict(tr['k_edge_attn_fwd'], fused_with_attention=True))
        for k, v in tr.items():
            if k in out:
                out[k]['traffic'] = int((2 * v['FETCH_SIZE_KB'] + v['WRITE_SIZE_KB']) * 1024)
                if v.get('fused_with_attention'):
                    out[k]['traffic_note'] = ('PMC bytes of k_edge_attn_fwd (edge messages + cross attention in one launch: what '
                                              'runs in the step at this size), against the standalone k_edge_fwd launch time')
                out[k]['traffic_stale'] = stale['stale']      # True: counters of another state of the kernel sources
                out[k]['traffic_source'] = (f'profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                            'separate passes; NOT measured in this run - hardware counters cannot be read '
                                            'from inside the process)')
                out[k]['hbm_measured_traffic_GBps'] = round(out[k]['traffic'] / (out[k]['avg_launch_us'] * 1e-6) / 1e9, 1)
                out[k]['hbm_frac_measured_traffic'] = round(out[k]['hbm_measured_traffic_GBps'] / PEAK_HBM_GBS, 4)
    except Exception:
        pass
    return out


def profile_step(compute, dev, reps=4):
    """Per-kernel durations of ONE eager step, measured live with the library's launch profiler (eqd_profile_*: a HIP
    event after every launch of the library, on the stream it launches on).  The step is enqueued behind a ~3 ms spin
    kernel so that the host stays ahead of the GPU and consecutive events bracket kernel execution, not host gaps; the
    cost of the event record itself is calibrated with empty intervals and subtracted.  Returns
    ({name: [calls per step, us per step]}, event overhead us, launches per step)."""
    from equidock_public_amd import _lib
    lib = _lib.load_library()
    stream = torch.cuda.current_stream(dev)
    st = _lib.stream_ptr(dev)
    compute()
    torch.cuda.synchronize()
    # empty-interval calibration: back-to-back event records behind a spin kernel
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    torch.cuda._sleep(4_000_000)
    for e in evs:
        e.record(stream)
    torch.cuda.synchronize()
    gaps = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(evs[:-1], evs[1:]))
    overhead = gaps[len(gaps) // 2]
    acc, n_launch = {}, 0
    for _ in range(reps):
        torch.cuda._sleep(8_000_000)        # ~3-4 ms: longer than the host needs to enqueue one eager step
        _lib.profiling = True
        _lib.check(lib.eqd_profile_begin(st, 1024))
        try:
            compute()
        finally:
            n = lib.eqd_profile_end()
            _lib.profiling = False
        if n < 0:
            raise RuntimeError('eqd_profile_end failed')
        n_launch = n
        for i in range(n):
            nm = lib.eqd_profile_name(i).decode()
            a = acc.setdefault(nm, [0, 0.0])
            a[0] += 1
            a[1] += max(0.0, lib.eqd_profile_us(i) - overhead)
    return {k: [v[0] / reps, v[1] / reps] for k, v in acc.items()}, overhead, n_launch


def kernel_rooflines(prof, work, bf16, step_us, pmc, traffic=None):
    """roofline_all: one entry per kernel family that takes >= 0.5 % of the step's kernel time (plus the two edge kernels
    always); `traffic` = per-family {FETCH_SIZE_KB, WRITE_SIZE_KB} per launch of the newest committed PMC summary."""
    out = {}
    total = sum(v[1] for v in prof.values())
    peak_tf = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
    for name, (calls, us) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        share = us / total if total > 0 else 0.0
        w = work.get(name)
        if w is None or (w['flops'] == 0 and w['bytes'] == 0) or (share < 0.005 and not name.startswith('k_edge')):
            continue
        t = us * 1e-6
        tf_x = w['flops'] / t / 1e12
        tf_w = w['flops_written'] / t / 1e12 if w['flops_written'] else None
        gb = w['bytes'] / t / 1e9
        e = {"calls_per_step": round(calls, 2), "us_per_step": round(us, 1), "avg_launch_us": round(us / calls, 2),
             "share_of_kernel_time": round(share, 4),
             "executed_flops_per_step": w['flops'], "executed_TFLOPs": round(tf_x, 2),
             "executed_frac_of_mfma_peak": round(tf_x / peak_tf, 4),
             "algorithmic_bytes_per_step": w['bytes'], "hbm_algorithmic_GBps": round(gb, 1),
             "hbm_frac_algorithmic": round(gb / PEAK_HBM_GBS, 4)}
        if tf_w is not None:
            e["as_written_flops_per_step"] = w['flops_written']
            e["as_written_TFLOPs"] = round(tf_w, 2)
            e["as_written_frac_of_mfma_peak"] = round(tf_w / peak_tf, 4)
        if w['flops'] == 0:
            e["bound"] = "hbm"
        else:
            e["bound"] = "mfma" if tf_x / peak_tf >= gb / PEAK_HBM_GBS else "hbm"
        tr = (traffic or {}).get(name)
        if tr and w['bytes'] > 0 and calls > 0:
            # measured HBM-side bytes per launch, (2 FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md: FETCH_SIZE counts
            # 128-B requests as 64 B on gfx950), over the algorithmic bytes per launch of the work model
            meas = (2 * tr['FETCH_SIZE_KB'] + tr['WRITE_SIZE_KB']) * 1024
            e["traffic_bytes_per_launch"] = int(meas)
            e["algorithmic_bytes_per_launch"] = int(w['bytes'] / calls)
            e["traffic_ratio"] = round(meas / (w['bytes'] / calls), 2)
        for k, v in pmc.items():        # hardware counters of an earlier rocprofv3 --pmc pass (profiles/*.json)
            if name in k and 'MfmaUtil' in v:
                e.setdefault("pmc", {})[k] = {"MfmaUtil_pct": v['MfmaUtil'],
                                              "executed_mfma_gflop_per_launch": v.get('executed_mfma_gflop_per_launch')}
        out[name] = e
    covered = sum(e["share_of_kernel_time"] for e in out.values())
    return out, total, covered


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(args_model, sd, pairs, budget_s=26.0):
    """The oracle (oracle/iegmn_port.py: the reference's op sequence, pinned to the reference's golden vectors) timed on
    the host cores, fwd + bwd of the fixed scalar loss, on a bounded sample of the workload's pairs.  torch's CPU
    parallelism does not scale on these op sizes (on a 128-core host all cores are ~10x SLOWER than one), so the batch
    step is timed at several thread counts and the BEST is reported:
      value        pairs/s of the sample batch in ONE step (the reference's DataLoader batch) at the best thread count
                   (`cores` = that count; `by_threads` = the whole table, including all physical cores);
      one_thread   the same step with torch.set_num_threads(1);
      b1_per_step  one pair per step at the best thread count - the reference's best batch size, because its dense
                   batch-wide attention mask makes the per-pair cost grow with the batch.
    Batches whose dense (sum n_lig x sum n_rec) mask would not be reasonable on the host (> 4000 x 4000) use the
    oracle's block-diagonal mode - equal to the reference's result, cheaper than its arithmetic - and say so."""
    from equidock_public_amd import graph
    from oracle import iegmn_port as port
    cores = physical_cores()
    prev = torch.get_num_threads()

    def run(ps, threads, budget, max_steps=3):
        torch.set_num_threads(threads)
        g = graph.batch_pairs(ps)
        raw = port.raw_from_graph(g)
        faithful = sum(raw['lig_counts']) * sum(raw['rec_counts']) <= 16_000_000
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

        def step():
            for v in leaves.values():
                v.grad = None
            port.scalar_loss(port.forward(leaves, args_model, raw, faithful=faithful)).backward()
        t0 = time.perf_counter()
        step()      # warm-up
        warm = time.perf_counter() - t0
        times, end = [], time.perf_counter() + max(0.0, budget - warm)
        while not times or (len(times) < max_steps and time.perf_counter() < end):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        times.sort()
        return len(ps) / times[len(times) // 2], len(times), faithful
    try:
        n = len(pairs)
        # bounded sample: at most 8 pairs of the batch; a single pair when the proteins are stress-sized
        sample = pairs[:min(n, 8)]
        if sum(len(l['x']) + len(r['x']) for l, r in sample) > 9000:
            sample = pairs[:1]
        counts = sorted({c for c in (1, 4, 16, 64, cores) if c <= cores})
        table, faithful = {}, True
        for c in counts:
            v, k, faithful = run(sample, c, budget_s * 0.8 / len(counts))
            table[c] = (v, k)
        best = max(table, key=lambda c: table[c][0])
        v_b1, k_b1, f1 = run(sample[:1], best, budget_s * 0.2)
    finally:
        torch.set_num_threads(prev)
    mode = "faithful mode (dense batch-wide mask)" if faithful else "block-diagonal mode (dense mask too large for the host)"
    return {"value": round(table[best][0], 3), "unit": "pairs/s", "cores": best, "kind": "port",
            "one_thread": round(table[1][0], 3), "b1_per_step": round(v_b1, 3), "physical_cores": cores,
            "by_threads": {str(c): round(v, 3) for c, (v, _) in table.items()},
            "sample": f"{len(sample)} of the workload's {n} pairs per GPU in one step, fwd+bwd, {mode}; 1 warm-up + "
                      f"{'/'.join(str(k) for _, k in table.values())} timed steps at {'/'.join(str(c) for c in table)} "
                      f"threads (host has {cores} physical cores; value = best: {best} threads); b1_per_step: {k_b1} steps of 1 "
                      f"pair at {best} threads ({'faithful' if f1 else 'block-diagonal'}); oracle/iegmn_port.py, torch "
                      f"{torch.__version__} CPU"}


def running_digest():
    from equidock_public_amd.build import csrc_digest
    return csrc_digest()


def staleness(counter_digest):
    """{"csrc_digest", "running_csrc_digest", "stale"} for a committed counter summary: the hardware counters are collected
    by separate rocprofv3 --pmc runs and committed under profiles/; they describe THIS library only when the kernel sources
    they were measured on (equidock_public_amd.build.csrc_digest, stamped by profiles/measure_r06.sh) are the ones running
    now.  A summary without a stamp (rounds 1-5) is stale by definition."""
    run = running_digest()
    return {"csrc_digest": counter_digest, "running_csrc_digest": run, "stale": counter_digest != run}


def load_traffic(workload):
    """Per-family FETCH_SIZE / WRITE_SIZE (KB per launch) of `workload` from the newest committed PMC summary, and where it
    came from (file, the workload's own collection stamp and git hash when the summary carries them)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*traffic.json')), key=_profile_order)      # newest last
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if workload in d:
            meta = d.get('_workloads', {}).get(workload, {})
            src = {"file": 'profiles/' + os.path.basename(f), "collected": meta.get('collected', d.get('_collected')),
                   "git_head": meta.get('git_head')}
            src.update(staleness(meta.get('csrc_digest')))
            return d[workload], src
    return {}, None


def load_issue_floor(workload):
    """Per-family issue floors of `workload` from the newest committed SQ-counter summary (profiles/*issue_floor.json,
    profiles/issue_floor.py): the busy time of the launch's busiest issue pipe (VALU / LDS / MFMA) and the launch time it was
    measured against - the roof a latency-bound kernel is actually running under."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, 'profiles', '*issue_floor.json')), key=_profile_order)):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if workload in d:
            return d[workload], dict({"file": 'profiles/' + os.path.basename(f)}, **staleness(d.get('_csrc_digest')))
    return {}, None


def load_pmc(workload):
    """MFMA utilisation / executed MFMA FLOPs per kernel from the newest committed PMC summary (profiles/*pmc_mfma*.json)."""
    import glob
    best = {}
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_mfma*.json')), key=_profile_order):
        try:
            full = json.load(open(f))
            d = full.get(workload)
            if d:
                stale = staleness(full.get('_workloads', {}).get(workload, {}).get('csrc_digest'))['stale']
                best = {k: dict(v, source=os.path.basename(f), stale=stale) for k, v in d.items()}
        except Exception:
            pass
    return best


class _Run:
    """One timed workload: everything main() needs to report it."""


def run_workload(workload, dtype, steps, warmup, dev, rank, world, use_dist, backend, dropout=0.0, dropout_masks='library',
                 eager=False, time_allreduce=True):
    """Build the model + the synthetic batch of `workload`, capture zero-grad -> forward -> loss -> backward (+ the RCCL
    all-reduce under a process group) into a hipGraph, run `warmup` untimed and `steps` timed steps bracketed by
    barrier + synchronize, max over ranks.  Returns a _Run (seconds for the timed steps in .dt)."""
    import torch.distributed as dist
    from equidock_public_amd import config, graph, model, parallel, synthetic
    from equidock_public_amd import losses
    R = _Run()
    ppg, uniform, L, shared, skh, wl_dtype, desc = WORKLOADS[workload]
    dtype = dtype or wl_dtype
    args_model = config.published_args(iegmn_n_lays=L, shared_layers=shared, skip_weight_h=skh, device=dev)
    if dtype == 'bf16':
        args_model['hip_storage_dtype'] = 'bf16'
    if dropout > 0:
        args_model['dropout'] = dropout
        args_model['hip_dropout_masks'] = dropout_masks
    sd = config.seeded_state_dict(args_model, seed=0)
    net = model.Rigid_Body_Docking_Net(args_model).to(dev)
    net.load_state_dict(sd)
    net.train(True)      # the step is a TRAINING step (forward + backward); with --dropout the masks are live
    if os.environ.get('EQD_BENCH_DROPOUT_PACK'):      # 'torch': the mask packing in torch operators (comparison runs)
        net.iegmn_original.dropout_pack = os.environ['EQD_BENCH_DROPOUT_PACK']
    sizes = [uniform] * ppg if uniform else synthetic.realistic_sizes(ppg, R_SIZE_SEED + rank)
    pairs = synthetic.make_pairs(sizes, seed=1000 + rank)
    g = graph.batch_pairs(pairs).to(dev)
    packed = g.pack()
    lig_w = torch.cat([torch.full((n, 1), 1.0 / (3 * n)) for n in packed.lig_counts]).to(dev)
    reducer = parallel.FlatGradAllReduce(net)
    scalar_loss = losses.ScalarLoss(packed, args_model['num_att_heads'])

    def compute():
        # zero grads -> forward -> fixed scalar loss (value + its gradient w.r.t. the outputs: ONE launch, eqd_scalar_loss)
        # -> backward of the model from those output gradients (== loss.backward())
        reducer.zero()
        lig, Yl, Yr, T, b = net.forward_batched(g)
        loss, grads = scalar_loss(lig, Yl, Yr)
        torch.autograd.backward([lig, Yl, Yr], list(grads))
        return loss

    def compute_torch_loss():      # the same step with the loss written in torch ops (cross-check of the fused kernel)
        reducer.zero()
        lig, Yl, Yr, T, b = net.forward_batched(g)
        loss = batched_loss(lig, Yl, Yr, lig_w)
        loss.backward()
        return loss
    torch.manual_seed(4242)     # (same dropout masks in both evaluations)
    l_ref = float(compute_torch_loss().detach())
    g_ref = reducer.flat.clone()
    torch.manual_seed(4242)
    l_got = float(compute())
    if abs(l_got - l_ref) > 1e-5 * abs(l_ref) or float((reducer.flat - g_ref).abs().max()) > 1e-5 * float(g_ref.abs().max()):
        raise SystemExit(f"fused scalar loss disagrees with its torch formulation: {l_got} vs {l_ref}")

    # Default: capture zero-grad -> forward -> loss -> backward ONCE into a hipGraph and replay it every step (every
    # kernel still runs every step).  With a process group the RCCL all-reduce of the flat gradient is captured INSIDE the
    # graph (RCCL supports stream capture: the collective becomes a graph node, no host enqueue per step); if that
    # capture fails the graph is rebuilt without it and the collective is enqueued from the host after every replay.
    # --eager launches everything from the host; a failed capture falls back to that as well.
    graph_mode = 'eager'
    static_loss = None
    cuda_graph = None
    allreduce_in_graph = False

    def capture(with_allreduce):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()      # no collective in flight while the stream is capturing
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                compute()
                if with_allreduce:
                    reducer.reduce(force=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode='thread_local'):
            loss_ = compute()
            if with_allreduce:
                reducer.reduce(force=True)
        return gr, loss_
    if not eager:
        want_ar = use_dist and backend == 'nccl' and os.environ.get('EQD_BENCH_ALLREDUCE_IN_GRAPH', '1') == '1'
        for with_ar in ([True, False] if want_ar else [False]):
            try:
                cuda_graph, static_loss = capture(with_ar)
                graph_mode = 'hipGraph replay'
                allreduce_in_graph = with_ar
                break
            except Exception as e:   # capture not possible: without the collective, then eager launches (still the HIP path)
                print(f"[bench] graph capture ({'with' if with_ar else 'without'} the all-reduce) failed "
                      f"({type(e).__name__}: {e})", file=sys.stderr)
                cuda_graph = None
                torch.cuda.synchronize()

    def step():
        if cuda_graph is not None:
            cuda_graph.replay()
            loss = static_loss
        else:
            loss = compute()
        if use_dist and not allreduce_in_graph:
            reducer.reduce(force=True)
        return loss

    for _ in range(warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    host_dt = time.perf_counter() - t0      # host-side enqueue time (== dt when the step is launch-bound)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the collective on its own (outside the timed region): HIP events around batches of back-to-back all-reduces of the
    # flat gradient buffer, every rank taking part; max over ranks
    allreduce_us = None
    if use_dist and time_allreduce:
        for _ in range(3):
            reducer.reduce(force=True)
        torch.cuda.synchronize()
        dist.barrier()
        reps, per = 5, 10
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record()
            for _ in range(per):
                reducer.reduce(force=True)
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) / per * 1e3 for e0, e1 in evs)
        tt = torch.tensor([ts[len(ts) // 2]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        allreduce_us = float(tt.item())
    status = net.iegmn_original.last_svd_status
    R.svd_bad = int((status != 0).sum().item())
    R.workload, R.dtype, R.desc, R.ppg, R.L, R.shared, R.skh = workload, dtype, desc, ppg, L, shared, skh
    R.uniform, R.sizes, R.pairs, R.sd, R.args_model = uniform, sizes, pairs, sd, args_model
    R.net, R.packed, R.reducer, R.compute, R.g = net, packed, reducer, compute, g
    R.dt, R.host_dt, R.loss, R.steps, R.warmup = dt, host_dt, float(loss.detach()), steps, warmup
    R.graph_mode, R.allreduce_in_graph, R.allreduce_us = graph_mode, allreduce_in_graph, allreduce_us
    R.launches = None
    return R


def time_inference(net, g, ppg, dev, steps=30, warmup=5):
    """Forward-only throughput (eval mode, torch.no_grad(): eqd_model_forward without a saved-state buffer), hipGraph replay."""
    was_training = net.training
    net.eval()
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    net.forward_batched(g)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode='thread_local'):
                outs = net.forward_batched(g)
            for _ in range(warmup):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gr.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        del outs
    finally:
        net.train(was_training)
    return {"metric": "protein-pairs/sec, forward only (inference)", "value": round(ppg * steps / dt, 2), "unit": "pairs/s",
            "ms_per_forward": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup, "launch_mode": "hipGraph replay"}


def time_train_step(R, dev, world, use_dist, backend, steps=20, warmup=5, n_pocket=30, eager_tail=True):
    """The reference's TRAINING step on this workload's batch (src/train.py:98-154): model -> MSE + pocket OT (exact EMD,
    solved on the host) + body intersection -> backward (-> all-reduce under a process group), through
    equidock_public_amd.train_step.TrainStep: three hipGraphs around the one host join.  Never the headline `value`
    (BASELINE.json's metric is the IEGMN forward + backward alone); `ot_exposed_ms` = the GPU-timeline gap the host solve and
    its two copies leave between the graphs.  Bound ligand / receptor coordinates: the pairs' own x; pocket points: a seeded
    subset of `n_pocket` residues per protein (DB5.5 pockets have a few dozen residues)."""
    import numpy as np
    from equidock_public_amd import train_step as TS
    rng = np.random.default_rng(77)
    lig_t = torch.cat([torch.from_numpy(p[0]['x']) for p in R.pairs])
    rec_t = torch.cat([torch.from_numpy(p[1]['x']) for p in R.pairs])
    pl, pr = [], []
    for lig, rec in R.pairs:
        n = min(n_pocket, lig['x'].shape[0], rec['x'].shape[0])
        pl.append(torch.from_numpy(lig['new_x'][rng.permutation(lig['x'].shape[0])[:n]].copy()))
        pr.append(torch.from_numpy(rec['x'][rng.permutation(rec['x'].shape[0])[:n]].copy()))
    ts = TS.TrainStep(R.net, R.g, lig_t, rec_t, pl, pr, reducer=R.reducer, allreduce=bool(use_dist and backend == 'nccl'))
    l_eager = float(ts.step_eager().detach())
    torch.cuda.synchronize()
    g_eager = R.reducer.flat.clone()
    if os.environ.get('EQD_TRAINSTEP_NO_GRAPHS') != '1':      # (experiment: every launch from the host)
        ts.capture()
    import torch.distributed as dist
    for _ in range(warmup):
        ts.step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ts.step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if abs(float(loss) - l_eager) > 1e-5 * abs(l_eager) or \
            (not use_dist and float((R.reducer.flat - g_eager).abs().max()) > 1e-5 * float(g_eager.abs().max())):
        raise RuntimeError(f"graph form of the training step disagrees with its autograd form: {float(loss)} vs {l_eager}")
    exposed, solve = [], []
    for _ in range(10):      # outside the timed region: the gap on the GPU timeline and the host solve alone
        ts.step()
        exposed.append(ts.last_ot_exposed_ms())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ts._solve_on_host()
        solve.append((time.perf_counter() - t1) * 1e3)
    dt_eager = float('nan')
    if eager_tail:      # (a profiler run leaves it out: the trace then ends with graph-form steps)
        t1 = time.perf_counter()
        for _ in range(5):
            ts.step_eager()
        torch.cuda.synchronize()
        dt_eager = (time.perf_counter() - t1) / 5
    ms = dt / steps * 1e3
    return {"metric": "protein-pairs/sec of the reference's training step (model fwd + MSE / pocket-OT / intersection losses + bwd)",
            "value": round(R.ppg * world * steps / dt, 2), "unit": "pairs/s", "ms_per_step": round(ms, 4), "steps": steps,
            "warmup": warmup, "loss": float(loss), "pocket_points_per_pair": n_pocket,
            "ot_exposed_ms": round(sorted(exposed)[len(exposed) // 2], 4),
            "host_solve_ms": round(sorted(solve)[len(solve) // 2], 4),
            "model_only_ms_per_step": round(R.dt / R.steps * 1e3, 4),
            "autograd_form_ms_per_step": (round(dt_eager * 1e3, 4) if eager_tail else None),
            "launch_mode": "three hipGraphs (forward + cost matrices | pair terms | OT terms + backward) around one host join",
            "weights": {"pocket_ot": ts.w_ot, "intersection": ts.w_int, "sigma": ts.sigma, "surface_ct": ts.ct}}


def north_star_hbm_entry(rl):
    """north_star's "achieved HBM bandwidth on the IEGMN message kernel": per edge kernel the live launch time of THIS run
    with (i) the algorithmic bytes and (ii) the rocprofv3-measured bytes of the committed PMC pass."""
    out = {}
    for k, v in rl.items():
        out[k] = {"avg_launch_us": v["avg_launch_us"], "algorithmic_bytes_per_launch": v["algorithmic_bytes_per_launch"],
                  "hbm_frac_algorithmic": v["hbm_frac_algorithmic"], "measured_bytes_per_launch": v.get("traffic"),
                  "hbm_frac_measured_traffic": v.get("hbm_frac_measured_traffic"),
                  "measured_traffic_stale": v.get("traffic_stale")}
    return out


def secondary_line(R, world):
    """The short form of a workload's result for the bench line's "secondary" block."""
    ms_step = R.dt / R.steps * 1e3
    flops_step = 3.0 * step_flops_as_written(R.sizes, R.L)
    peak_tf = PEAK_BF16_TFLOPS if R.dtype == 'bf16' else PEAK_FP32_TFLOPS
    out = {"workload": R.desc, "value": round(R.ppg * world * R.steps / R.dt, 2), "unit": "pairs/s",
           "ms_per_step": round(ms_step, 4), "steps": R.steps, "warmup": R.warmup,
           "dtype": "f32" if R.dtype == 'f32' else "bf16 GEMM inputs, fp32 accumulate",
           "pairs_per_gpu": R.ppg, "nodes_per_gpu": R.packed.n_nodes, "edges_per_gpu": R.packed.n_edges,
           "launch_mode": R.graph_mode, "loss": R.loss, "svd_guard_pairs": R.svd_bad,
           "whole_step_frac_of_mfma_peak": round(flops_step / (ms_step * 1e-3) / 1e12 / peak_tf, 4)}
    if not R.uniform:
        pp = sum(x * y for x, y in R.sizes)
        out["residue_pairs_per_s"] = round(pp * world * R.steps / R.dt, 1)
    if R.allreduce_us is not None:
        out["allreduce_in_graph"] = bool(R.allreduce_in_graph)
        out["allreduce_us_per_step"] = round(R.allreduce_us, 2)
    return out


def relaunch_under_torchrun(n):
    """Run this command line again as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same
    arguments>` on a free local port; stdout / stderr pass through (rank 0 prints the JSON line), the launcher's exit code
    is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] --gpus %d without a launcher: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help="default B (BASELINE.json configs[1]) WITH the \"secondary\" / \"inference\" blocks; an explicitly "
                         "named workload (also `--workload B`) runs that workload alone, so that a profiler around the "
                         "command sees one workload's kernels only")
    ap.add_argument('--dtype', default=None, choices=('f32', 'bf16'),
                    help="arithmetic of the GEMMs: f32 (default for A, B, C, E) or bf16 inputs with fp32 accumulate "
                         "(default for D; `--workload C --dtype bf16` is BASELINE.json configs[2])")
    ap.add_argument('--dropout', type=float, default=0.0,
                    help="args['dropout'] of the model, in training mode (src/utils/args.py:240 draws 0 or 0.25): every step "
                         "then draws fresh nn.Dropout masks with torch's device generator in the reference's order (inside "
                         "the replayed graph) and the kernels apply them; reported WITHOUT the roofline / CPU-baseline parts")
    ap.add_argument('--dropout-masks', default='library', choices=('torch', 'library'),
                    help="with --dropout: 'torch' = nn.Dropout's own random stream (torch's dropout on [E, 64] tensors of "
                         "ones, bit-packed by eqd_dropout_pack_edges); 'library' = eqd_dropout_draw (counter-based, one launch, "
                         "no [E, 64] tensors; args['hip_dropout_masks'])")
    ap.add_argument('--train-step', action='store_true',
                    help="run ONLY the reference's training step (model + MSE / pocket-OT / intersection losses + backward: "
                         "train_step.TrainStep, three hipGraphs around the exact-OT host solve) on the named workload and print its "
                         "block - what a profiler should wrap to see the gap the host solve leaves on the GPU timeline")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help="skip the \"secondary\" block of the default (workload B) line: short timed runs of the other BASELINE "
                         "configs - C in bf16 (configs[2]), E (configs[4]) and the ragged workload R on one GPU; D (configs[3], "
                         "64 x (300,300) per GPU in bf16 with the RCCL all-reduce) under torch.distributed.run")
    ap.add_argument('--eager', action='store_true', help='launch every kernel of every step from the host instead of replaying a captured hipGraph of the step (zero-grad, forward, loss, backward; the gradient all-reduce always runs outside the graph).  Same kernels either way; the replay takes the host (torch autograd + the launches, 0.7-1.5 ms depending on the box) off the critical path of a 1.5 ms step')
    ap.add_argument('--graph', action='store_true', help='(default; kept for older command lines)')
    a = ap.parse_args()
    if a.workload is not None:      # a named workload is measured alone (profiles: one workload per kernel table)
        a.no_secondary = True
    else:
        a.workload = 'B'

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU (the
        # driver's N > 1 command line is the launcher form; this makes the plain form give the same single JSON line)
        raise SystemExit(relaunch_under_torchrun(a.gpus))
    if a.gpus != world:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)
    import torch.distributed as dist
    if os.environ.get('EQD_BENCH_DRY_RUN') == '1':
        # launcher rehearsal for boxes without a GPU (tests/test_data_parallel.py): rendezvous over gloo, the barrier +
        # max-over-ranks reduction of the timing, ONE line from rank 0, clean shutdown - no workload, not a measurement
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if 'RANK' in os.environ:
            dist.init_process_group('gloo')
            dist.barrier()
            tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.barrier()
            dist.destroy_process_group()
            assert int(tt.item()) == world
            if os.environ.get('EQD_BENCH_DRY_RUN_FAIL') == '1' and rank == world - 1:
                raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "workload": a.workload}),
                  flush=True)
        return
    # EQD_BENCH_ONE_DEVICE=1 + EQD_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow (barriers, max over ranks,
    # rank-0 report, gradient all-reduce) on a box with a single GPU; real runs use one GPU per rank over RCCL
    one_dev = os.environ.get('EQD_BENCH_ONE_DEVICE') == '1'
    dev = torch.device('cuda', 0 if one_dev else local_rank)
    torch.cuda.set_device(dev)
    # under torch.distributed.run (RANK set) the process group is created even for a world of one, so that the RCCL
    # all-reduce of the flat gradient is part of the step and gets measured; a plain `python bench.py` has no group
    use_dist = world > 1 or 'RANK' in os.environ
    backend = None
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        backend = os.environ.get('EQD_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    R = run_workload(a.workload, a.dtype, a.steps, a.warmup, dev, rank, world, use_dist, backend, dropout=a.dropout,
                     dropout_masks=a.dropout_masks, eager=a.eager)
    if a.train_step:
        blk = time_train_step(R, dev, world, use_dist, backend, steps=a.steps, warmup=a.warmup,
                              eager_tail=os.environ.get('EQD_BENCH_TRAIN_EAGER_TAIL', '1') != '0')
        if rank == 0:
            print(json.dumps(dict(blk, workload=R.desc, dtype=R.dtype)), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.dropout > 0:
        a.no_cpu_baseline = a.no_roofline = True
    dtype, desc, ppg, L, shared, skh = R.dtype, R.desc, R.ppg, R.L, R.shared, R.skh
    uniform, sizes, pairs, sd, net, packed, reducer, compute = R.uniform, R.sizes, R.pairs, R.sd, R.net, R.packed, R.reducer, R.compute
    dt, host_dt, graph_mode, allreduce_in_graph, allreduce_us = R.dt, R.host_dt, R.graph_mode, R.allreduce_in_graph, R.allreduce_us
    svd_bad = R.svd_bad
    # the other BASELINE configs, timed the same way (graph replay, barrier + synchronize on both sides, max over ranks) with
    # fewer steps, so that the driver's default run carries a number for each of them
    secondary, ns_hbm = {}, {}
    if a.workload == 'B' and a.dropout == 0 and not a.eager and not a.no_secondary:
        todo = [('D', 'D', None)] if use_dist else [('C_bf16', 'C', 'bf16'), ('C_f32', 'C', 'f32'), ('E', 'E', None),
                                                    ('R', 'R', None)]
        for key, wl, dt_ in todo:
            t_sec = time.perf_counter()
            try:
                R2 = run_workload(wl, dt_, 10, 3, dev, rank, world, use_dist, backend)
                secondary[key] = secondary_line(R2, world)
                if rank == 0 and not a.no_roofline and R2.uniform:
                    # the message kernels of this workload, timed standalone like the primary "roofline" (north_star_hbm)
                    ns_hbm[key] = north_star_hbm_entry(edge_kernel_rooflines(R2.net, R2.packed, dev, wl,
                                                                             bf16=(R2.dtype == 'bf16')))
                if key in ('C_bf16', 'D'):      # the reference's full training step on the bf16 configuration
                    try:
                        secondary['train_step_' + key] = time_train_step(R2, dev, world, use_dist, backend, steps=10, warmup=3)
                    except Exception as e:
                        secondary['train_step_' + key] = {"error": f"{type(e).__name__}: {e}"}
                secondary[key]["wall_s_incl_setup"] = round(time.perf_counter() - t_sec, 2)
                del R2
            except Exception as e:      # the primary line must not die on a secondary workload
                secondary[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    if a.workload == 'B' and a.dropout == 0 and not a.eager and not a.no_secondary:
        try:
            secondary['train_step_B'] = time_train_step(R, dev, world, use_dist, backend)
        except Exception as e:
            secondary['train_step_B'] = {"error": f"{type(e).__name__}: {e}"}
    # inference (src/inference_rigid.py:194: the forward alone, no state kept): the primary workload's batch under
    # torch.no_grad(), replayed from its own hipGraph, timed like the step
    inference = None
    if a.workload == 'B' and a.dropout == 0 and not a.eager and not a.no_secondary and world == 1:
        try:
            inference = time_inference(R.net, R.g, ppg, dev)
        except Exception as e:
            inference = {"error": f"{type(e).__name__}: {e}"}
    out = None
    if rank == 0:
        total_pairs = ppg * world * a.steps
        ms_step = dt / a.steps * 1e3
        out = {
            "metric": "protein-pairs/sec (fwd+bwd) per IEGMN stack", "value": round(total_pairs / dt, 2),
            "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if dtype == 'f32' else "bf16 (GEMM inputs; fp32 accumulate, fp32 coordinates / statistics / softmax / Kabsch)",
            "data": "synthetic",
            "config": {"workload": desc, "pairs_per_gpu": ppg, "nodes_per_gpu": packed.n_nodes,
                       "edges_per_gpu": packed.n_edges, "layers": L, "parallelism": f"dp{world}",
                       "weights": "PyTorch default init (seed 0), ROT key/query x40 (SURVEY.md section 8c)",
                       "loss": R.loss, "svd_guard_pairs": svd_bad,
                       "host_enqueue_ms_per_step": round(host_dt / a.steps * 1e3, 4), "launch_mode": graph_mode,
                       "dropout": a.dropout, "dropout_masks": (a.dropout_masks if a.dropout > 0 else None)},
            # data parallel: ranks of the RCCL communicator the flat-gradient all-reduce ran on (0 = no process group: a
            # plain single-process run), whether the collective was a node of the replayed hipGraph, and its duration
            # alone (HIP events around back-to-back all-reduces of the buffer, after the timed region; max over ranks)
            "rccl_ranks": (world if (use_dist and backend == 'nccl') else 0),
            "allreduce_in_graph": bool(allreduce_in_graph),
            "allreduce_us_per_step": (None if allreduce_us is None else round(allreduce_us, 2)),
            "allreduce_bytes": int(reducer.flat.numel() * 4),
        }
        if not uniform:      # ragged workload: the size spread and the quadratic-term rate
            nls, nrs = [s[0] for s in sizes], [s[1] for s in sizes]
            pp = sum(x * y for x, y in sizes)
            out["config"].update({
                "sizes": {"ligand_min_median_max": [min(nls), sorted(nls)[len(nls) // 2], max(nls)],
                          "receptor_min_median_max": [min(nrs), sorted(nrs)[len(nrs) // 2], max(nrs)],
                          "sum_nlig_x_nrec": pp, "size_seed": R_SIZE_SEED,
                          "largest_pair_share_of_sum_nlig_x_nrec": round(max(x * y for x, y in sizes) / pp, 4)}})
            out["residue_pairs_per_s"] = round(pp * world * a.steps / dt, 1)
        flops_step = 3.0 * step_flops_as_written(sizes, L)
        peak_tf = PEAK_BF16_TFLOPS if dtype == 'bf16' else PEAK_FP32_TFLOPS
        out["whole_step"] = {"as_written_flops_per_step": flops_step,
                             "achieved_TFLOPs": round(flops_step / (ms_step * 1e-3) / 1e12, 2), "peak_TFLOPs": peak_tf,
                             "frac": round(flops_step / (ms_step * 1e-3) / 1e12 / peak_tf, 4),
                             "note": "SURVEY.md section 8d FLOPs of the model as written, fwd + bwd = 3x fwd, over ms_per_step"}
        if not a.no_roofline:
            rl = edge_kernel_rooflines(net, packed, dev, a.workload, bf16=(dtype == 'bf16'))
            dom = max(rl, key=lambda k: rl[k]["avg_launch_us"])
            out["roofline"] = dict(rl[dom], kernel=dom)
            out["north_star_hbm"] = dict({a.workload + ('_bf16' if dtype == 'bf16' else ''): north_star_hbm_entry(rl)}, **ns_hbm)
            out["north_star_hbm"]["note"] = ("north_star: >= 50 % achieved HBM bandwidth on the IEGMN message kernel.  Per edge "
                                             "kernel and workload: launch time of this run (HIP events, standalone launches) "
                                             "against algorithmic bytes and against the rocprofv3-measured bytes of the newest "
                                             "committed PMC pass (profiles/*traffic.json), as fractions of 8 TB/s.  The fused "
                                             "fp32 kernel is MFMA-bound (SURVEY.md section 8d: 230 FLOP/B against a ridge of "
                                             "20), so neither fraction can reach 0.5 there")
            try:
                prof, ev_us, n_launch = profile_step(compute, dev)
                work = kernel_work_model(sizes, L, packed.n_nodes, packed.n_edges, fused_fwd='k_edge_attn_fwd' in prof,
                                         rowwave=next((k for k in ('k_rowres', 'k_rowwave') if k in prof), None),
                                         fused_gather='k_attn_bwd_gather' in prof, ds_handoff='k_attn_bwd_kvds' in prof,
                                         wide_resident=(dtype == 'bf16' and 'k_rowres' in prof and 'k_rowchain' not in prof))
                wkey = a.workload + ('_bf16' if dtype == 'bf16' else '')
                traffic, traffic_src = load_traffic(wkey)
                allk, ktot, covered = kernel_rooflines(prof, work, dtype == 'bf16', ms_step * 1e3, load_pmc(wkey), traffic)
                floors, floor_src = load_issue_floor(wkey)
                for k, e in allk.items():      # issue floor of the family (SQ counters of an earlier rocprofv3 --pmc pass)
                    fl = floors.get(k) or next((v for kk, v in floors.items() if k.startswith(kk)), None)
                    if fl:
                        e["issue_floor"] = dict(fl, source=floor_src["file"], stale=floor_src["stale"],
                                                live_frac_of_floor=round(fl["floor_us"] / e["avg_launch_us"], 4))
                for k in ('k_edge_fwd', 'k_edge_bwd'):     # keep the standalone batched-launch figures beside the in-step ones
                    if k in allk:
                        allk[k]["standalone"] = rl[k]
                out["roofline_all"] = allk
                # hardware counters come from committed rocprofv3 --pmc summaries: say which, and whether they were collected
                # on the kernel sources that are running now (a duration divided by another code
                # state's floor is not a measurement)
                out["counters"] = {"traffic": traffic_src, "issue_floor": floor_src,
                                   "stale": bool((traffic_src or {}).get("stale", True) or (floor_src or {}).get("stale", True))}
                out["step_profile"] = {
                    "library_launches_per_step": n_launch, "kernel_us_per_step": round(ktot, 1),
                    "share_of_kernel_time_in_roofline_all": round(covered, 4), "traffic_source": traffic_src,
                    "event_overhead_us_subtracted_per_launch": round(ev_us, 2),
                    "us_per_step_by_kernel": {k: round(v[1], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                    "method": "eqd_profile_* (HIP event after every library launch on the launch stream, eager step "
                              "enqueued behind a spin kernel), mean of 4 steps"}
            except Exception as e:      # the bench line must not die on the diagnostic part
                out["roofline_all"] = rl
                out["step_profile"] = {"error": f"{type(e).__name__}: {e}"}
        if secondary:
            out["secondary"] = secondary
        if inference is not None:
            out["inference"] = inference
        if world == 1 and not a.no_cpu_baseline:
            from equidock_public_amd import config
            out["cpu_baseline"] = cpu_baseline(config.published_args(iegmn_n_lays=L, shared_layers=shared,
                                                                     skip_weight_h=skh), sd, pairs)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
