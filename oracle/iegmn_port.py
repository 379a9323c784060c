"""ORACLE (test infrastructure, "port" kind): plain-PyTorch fp32 CPU restatement of the
reference hot path -- Rigid_Body_Docking_Net.forward and everything under it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product package `equidock_public_amd` never does; it fails loudly without its HIP library.

Parity pin: `oracle/make_golden.py` runs the REAL reference module
(/root/reference/src/model/rigid_docking_model.py, imported unmodified with the DGL stand-in in
oracle/_dgl_standin) in this container and commits its inputs/outputs/gradients under
tests/golden/; tests/test_oracle_golden.py checks this restatement against those vectors
(<= 1e-5 fp32).  The reference ships no tests of its own for this path, so
the golden vectors generated from the imported reference are the only pin.

Every function cites the reference lines it restates (paths relative to /root/reference).
"faithful=True" keeps the reference's op sequence including the dense batch-wide attention
mask (that is the CPU baseline that gets timed); "faithful=False" uses the per-pair
block-diagonal softmax the HIP path implements (equal to 9.5e-7).
"""
import math

import torch
import torch.nn.functional as F

RBF_SIGMAS = [1.5 ** k for k in range(15)]   # src/model/rigid_docking_model.py:116


def default_args(**over):
    """The published configuration family (src/utils/args.py:227-280, src/inference_rigid.py:90,93)."""
    a = dict(input_edge_feats_dim=27, dropout=0.0, nonlin='lkyrelu', cross_msgs=True, layer_norm='LN',
             layer_norm_coors='0', final_h_layer_norm='0', use_dist_in_layers=True, skip_weight_h=0.75,
             x_connection_init=0.0, leakyrelu_neg_slope=0.01, debug=False, device=torch.device('cpu'),
             graph_nodes='residues', rot_model='kb_att', noise_decay_rate=0.0, noise_initial=0.0,
             use_edge_features_in_gmn=True, use_mean_node_features=True, residue_emb_dim=64,
             iegmn_lay_hid_dim=64, shared_layers=False, num_att_heads=50, iegmn_n_lays=8, fine_tune=False)
    a.update(over)
    return a


class Kink:
    """Test aid for comparing gradients across fp32 summation orders.  A LeakyReLU pre-activation within rounding of 0
    takes the other slope (x100) under ANY reordering of the sums that produce it, and the parameter gradients that
    pass through it move by O(1 %) - in the reference itself as much as in any re-implementation.  `mode`:
      None    plain LeakyReLU (the reference's arithmetic);
      'count' plain LeakyReLU, and `near` accumulates how many pre-activations satisfy |z| <= eps * max(1, max|z|);
      'pos' / 'neg'  those pre-activations take the positive-side (1) / negative-side (slope) derivative.  The forward
              values move by at most eps, i.e. below fp32 resolution of the outputs.
      'given' every LeakyReLU takes the slope recorded in `given[tag]` (a bool tensor of the pre-activation's shape, True =
              derivative 1): the branch decisions of ANOTHER evaluation of the same function (the HIP library exports
              its own, eqd_model_lrelu_signs).  The gradient of that evaluation is then ONE well-defined object that can be
              compared plainly.  Where a given decision differs from this evaluation's own sign, |z| / max|z| is recorded
              in `flips` (tag, count, largest relative |z|): a legitimate difference is at rounding level (<~1e-6), anything
              larger is a bug in the other implementation and the tests assert on it.
    tests/parity_common.py compares gradients in 'given' mode; the three-evaluation hull ('count' / 'pos' / 'neg') is
    kept as a diagnostic (its width is reported) and for the golden vectors, whose reference gradient has the
    reference's own decisions baked in."""
    mode = None
    eps = 3e-7
    near = 0
    given = None
    flips = None
    flip_rels = None      # when a list: |z| / max|z| of EVERY differing decision (one tensor per tag), for distribution checks
    decisions = 0         # LeakyReLU decisions evaluated in 'given' mode (the denominator of a flip rate)


def _lrelu(x, slope, tag=None):
    y = F.leaky_relu(x, negative_slope=slope)
    if Kink.mode is None:
        return y
    if Kink.mode == 'given':
        pos = Kink.given[tag]
        if pos is None:         # recorded as "not evaluated by the other implementation" (e.g. q / k without cross_msgs)
            return y
        assert pos.shape == x.shape, (tag, pos.shape, x.shape)
        with torch.no_grad():
            diff = pos != (x > 0)
            n = int(diff.sum())
            Kink.decisions += x.numel()
            if n:
                zmax = max(float(x.abs().max()), 1e-30)
                Kink.flips.append((tag, n, float(x[diff].abs().max()) / zmax))
                if Kink.flip_rels is not None:
                    Kink.flip_rels.append((x[diff].abs() / zmax).reshape(-1))
        if n == 0:
            return y
        return torch.where(diff, x * torch.where(pos, 1.0, slope), y)
    with torch.no_grad():
        near = x.abs() <= Kink.eps * max(1.0, float(x.abs().max()) if x.numel() else 1.0)
        n = int(near.sum())
    Kink.near += n
    if Kink.mode == 'count' or n == 0:
        return y
    return torch.where(near, x * (1.0 if Kink.mode == 'pos' else slope), y)


class Bf16Mode:
    """Rounding points of the HIP path's bf16 mode (`hip_storage_dtype='bf16'`), restated so that the
    whole model can be held to a tight tolerance in that mode too: the INPUTS of every GEMM of the IEGMN layers are rounded
    to bf16 (round to nearest even), products are accumulated in fp32, everything else (coordinates, RBFs before the GEMM,
    biases, LeakyReLU, LayerNorm, softmax, residuals, the keypoint head and Kabsch) stays fp32 - i.e. the arithmetic of
    v_mfma_f32_16x16x16_bf16.  Rounded: the edge MLPs (first Linear split into node terms P[src] + Q[dst] plus a GEMM
    over [he, rbf]; W2 and Wc1 GEMMs), the node-level Linears (P / Q, att_mlp_Q / K / V, node_mlp.0 / .4,
    mlp_h_mean_ROT) and the attention contractions (q k^T and softmax-weights x v).  Rounding is the identity for
    autograd (the kernels' backward GEMMs round their own operands, which a CPU autograd cannot mirror: gradients are
    compared at bf16 resolution).  Not a reference option."""
    on = False


def rb16(t):
    """bf16 rounding of a GEMM input, identity for autograd"""
    return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()


def _edge_mlps_bf16(sd, pfx, slope, h, src, dst, he, rbf, side=''):
    """edge_mlp + coors_mlp of one edge type with the bf16 mode's rounding points (see Bf16Mode)."""
    W1, b1 = sd[pfx + 'edge_mlp.0.weight'], sd[pfx + 'edge_mlp.0.bias']
    d = h.shape[1]
    Pn = F.linear(rb16(h), rb16(W1[:, :d]))
    Qn = F.linear(rb16(h), rb16(W1[:, d:2 * d]), b1)
    z1 = Pn[src] + Qn[dst] + rb16(torch.cat([he, rbf], 1)) @ rb16(W1[:, 2 * d:]).t()
    a1 = F.layer_norm(_lrelu(z1, slope, pfx + 'edge_mlp.' + side), (z1.shape[1],), sd[pfx + 'edge_mlp.3.weight'],
                      sd[pfx + 'edge_mlp.3.bias'], 1e-5)
    msg = rb16(a1) @ rb16(sd[pfx + 'edge_mlp.4.weight']).t() + sd[pfx + 'edge_mlp.4.bias']
    ch = rb16(msg) @ rb16(sd[pfx + 'coors_mlp.0.weight']).t() + sd[pfx + 'coors_mlp.0.bias']
    coef = F.linear(_lrelu(ch, slope, pfx + 'coors_mlp.' + side), sd[pfx + 'coors_mlp.4.weight'], sd[pfx + 'coors_mlp.4.bias'])
    return msg, coef


def _lin(x, W, b=None):
    """nn.Linear; in Bf16Mode its two GEMM inputs are rounded to bf16."""
    if Bf16Mode.on:
        return F.linear(rb16(x), rb16(W), b)
    return F.linear(x, W, b)


def _mlp5(x, sd, prefix, slope, norm, side=''):
    """Linear -> Dropout(p=0) -> LeakyReLU -> LayerNorm/Identity -> Linear
    (edge_mlp :119-125, node_mlp :142-148, coors_mlp :153-159)."""
    y = _lin(x, sd[prefix + '.0.weight'], sd[prefix + '.0.bias'])
    y = _lrelu(y, slope, prefix + '.' + side)
    if norm == 'LN':
        y = F.layer_norm(y, (y.shape[-1],), sd[prefix + '.3.weight'], sd[prefix + '.3.bias'], 1e-5)
    else:
        assert norm == '0'
    return _lin(y, sd[prefix + '.4.weight'], sd[prefix + '.4.bias'])


def get_mask(lig_counts, rec_counts):
    """src/model/rigid_docking_model.py:68-78."""
    mask = torch.zeros(sum(lig_counts), sum(rec_counts))
    pl = pr = 0
    for ln, rn in zip(lig_counts, rec_counts):
        mask[pl:pl + ln, pr:pr + rn] = 1
        pl += ln
        pr += rn
    return mask


def _softmax_v_bf16(a, v):
    """softmax(a) @ v with the bf16 mode's rounding point: the un-normalised weights 2^(a log2 e - M), M = ceil of the row
    maximum in log2 units (an integer, so that any power-of-two rescaling - the kernels' online softmax - rounds to the same
    bf16 values), and v are rounded; the normaliser is the fp32 sum of the UNROUNDED weights."""
    a2 = a * 1.44269504088896341
    M = torch.ceil(a2.max(dim=1, keepdim=True).values).detach()
    e = torch.exp2(a2 - M)
    return torch.mm(rb16(e), rb16(v)) / e.sum(dim=1, keepdim=True)


def cross_attention_dense(q, k, v, mask):
    """src/model/rigid_docking_model.py:46-64 (no 1/sqrt(d), single head, -1000 fill)."""
    if Bf16Mode.on:
        a = mask * torch.mm(rb16(q), rb16(k).t()) - 1000. * (1. - mask)
        return _softmax_v_bf16(a, v)
    a = mask * torch.mm(q, k.t()) - 1000. * (1. - mask)
    return torch.mm(torch.softmax(a, dim=1), v)


def cross_attention_blockdiag(q, k, v, q_counts, k_counts):
    """Per-pair softmax: what the reference computes whenever in-pair logits are not < -900."""
    outs, qo, ko = [], 0, 0
    for nq, nk in zip(q_counts, k_counts):
        if Bf16Mode.on:
            a = torch.mm(rb16(q[qo:qo + nq]), rb16(k[ko:ko + nk]).t())
            outs.append(_softmax_v_bf16(a, v[ko:ko + nk]))
        else:
            a = torch.mm(q[qo:qo + nq], k[ko:ko + nk].t())
            outs.append(torch.mm(torch.softmax(a, dim=1), v[ko:ko + nk]))
        qo += nq
        ko += nk
    return torch.cat(outs, 0)


def _mean_by_dst(val, dst, n):
    """DGL update_all(copy_edge, mean): per-destination mean, zeros for in-degree 0
    (src/model/rigid_docking_model.py:274-283)."""
    acc = torch.zeros((n,) + tuple(val.shape[1:]), dtype=val.dtype).index_add(0, dst, val)
    deg = torch.zeros(n, dtype=val.dtype).index_add(0, dst, torch.ones(dst.numel(), dtype=val.dtype))
    return acc / deg.clamp(min=1.0).view(n, *([1] * (val.dim() - 1)))


def iegmn_layer(sd, pfx, args, d_in, raw, x_l, h_l, h0_l, he_l, x0_l, x_r, h_r, h0_r, he_r, x0_r, faithful, inter=None):
    """IEGMN_Layer.forward, src/model/rigid_docking_model.py:189-352."""
    slope = args['leakyrelu_neg_slope']
    out = {}
    msgs = {}
    for side, x, h, he, (src, dst) in (('l', x_l, h_l, he_l, (raw['ll_src'], raw['ll_dst'])),
                                       ('r', x_r, h_r, he_r, (raw['rr_src'], raw['rr_dst']))):
        src, dst = src.long(), dst.long()
        x_rel = x[src] - x[dst]                                              # :204-205
        d2 = (x_rel ** 2).sum(1, keepdim=True)                               # :208-209
        rbf = torch.cat([torch.exp(-d2 / s) for s in RBF_SIGMAS], dim=-1)    # :210
        if not args['use_dist_in_layers']:
            rbf = rbf * 0.                                                   # :216-218
        if Bf16Mode.on:
            msg, coef = _edge_mlps_bf16(sd, pfx, slope, h, src, dst, he, rbf, side)
        else:
            cat = torch.cat([h[src], h[dst], he, rbf], dim=-1)               # :226-234
            msg = _mlp5(cat, sd, pfx + 'edge_mlp', slope, args['layer_norm'], side)    # :236-237
            coef = _mlp5(msg, sd, pfx + 'coors_mlp', slope, args['layer_norm_coors'], side)   # :263-265
        n = x.shape[0]
        msgs[side] = dict(x_update=_mean_by_dst(x_rel * coef, dst, n),       # :274-277
                          aggr_msg=_mean_by_dst(msg, dst, n))                # :280-283

    def qkv(h, side):
        q = _lrelu(_lin(h, sd[pfx + 'att_mlp_Q.0.weight']), slope, pfx + 'att_mlp_Q.' + side)   # :130-133
        k = _lrelu(_lin(h, sd[pfx + 'att_mlp_K.0.weight']), slope, pfx + 'att_mlp_K.' + side)   # :134-137
        v = _lin(h, sd[pfx + 'att_mlp_V.0.weight'])                          # :138-140
        return q, k, v
    ql, kl, vl = qkv(h_l, 'l')
    qr, kr, vr = qkv(h_r, 'r')
    lc, rc = raw['lig_counts'], raw['rec_counts']
    if not args['cross_msgs']:
        cross_l, cross_r = ql * 0., qr * 0.                                  # :59-60
    elif faithful:
        mask = get_mask(lc, rc)                                              # :244
        cross_l = cross_attention_dense(ql, kr, vr, mask)                    # :247-251
        cross_r = cross_attention_dense(qr, kl, vl, mask.t())                # :252-256
    else:
        cross_l = cross_attention_blockdiag(ql, kr, vr, lc, rc)
        cross_r = cross_attention_blockdiag(qr, kl, vl, rc, lc)

    eta = args['x_connection_init']
    res = []
    for side, x, h, h0, x0, cross in (('l', x_l, h_l, h0_l, x0_l, cross_l), ('r', x_r, h_r, h0_r, x0_r, cross_r)):
        x_new = eta * x0 + (1. - eta) * x + msgs[side]['x_update']           # :286-292
        inp = torch.cat([h, msgs[side]['aggr_msg'], cross, h0], dim=-1)      # :319-329
        upd = _mlp5(inp, sd, pfx + 'node_mlp', slope, args['layer_norm'], side)
        if d_in == args['iegmn_lay_hid_dim']:                                # :332-337
            upd = args['skip_weight_h'] * upd + (1. - args['skip_weight_h']) * h
        assert args['final_h_layer_norm'] == '0'                             # :348-349 (Identity)
        res.append((x_new, upd))
    if inter is not None:
        inter['aggr_cross_l'], inter['aggr_cross_r'] = cross_l, cross_r
        inter['aggr_msg_l'], inter['aggr_msg_r'] = msgs['l']['aggr_msg'], msgs['r']['aggr_msg']
    return res[0][0], res[0][1], res[1][0], res[1][1]


def kabsch(Yr, Yl, rand_fn=None, status=None):
    """src/model/rigid_docking_model.py:563-589. Returns T (3,3), b (1,3), A (3,3)."""
    Yr_mean = Yr.mean(0, keepdim=True)
    Yl_mean = Yl.mean(0, keepdim=True)
    A = (Yr - Yr_mean).t() @ (Yl - Yl_mean)                                  # :567
    assert not torch.isnan(A).any()                                          # :570
    U, S, Vt = torch.linalg.svd(A)                                           # :571
    num_it = 0
    while torch.min(S) < 1e-3 or \
            torch.min(torch.abs((S ** 2).view(1, 3) - (S ** 2).view(3, 1) + torch.eye(3, dtype=S.dtype))) < 1e-2:   # :574
        draw = torch.rand(3, 3) if rand_fn is None else rand_fn(num_it)
        A = A + draw.to(A.dtype) * torch.eye(3, dtype=A.dtype)                  # :578
        U, S, Vt = torch.linalg.svd(A)
        num_it += 1
        if num_it > 10:
            raise RuntimeError('SVD consistently numerically unstable')      # :582-584 (sys.exit there)
    if status is not None:
        status.append(num_it)
    corr = torch.diag(torch.tensor([1., 1., float(torch.sign(torch.det(A.detach())))], dtype=A.dtype))   # :586
    T = (U @ corr) @ Vt                                                      # :587
    b = Yr_mean - torch.t(T @ Yl_mean.t())                                   # :589
    return T, b, A


def head(sd, args, raw, h_l, x_l, h_r, x_r, prefix='iegmn_original.', rand_fn=None):
    """The keypoint / Kabsch head of IEGMN.forward (:521-600) + the rigid apply of Rigid_Body_Docking_Net.forward (:665) on a
    given last-layer state (h, x of the ligand and the receptor nodes of the whole batch).  forward() ends with it; the
    tests also call it on the HIP library's own (h_L, x_L) to compare the head's backward on its own.
    Returns (ligs, Yls, Yrs, Ts, bs, As, svd_iterations)."""
    slope = args['leakyrelu_neg_slope']
    K = args['num_att_heads']
    d = args['iegmn_lay_hid_dim']
    Wk = sd[prefix + 'att_mlp_key_ROT.0.weight']
    Wq = sd[prefix + 'att_mlp_query_ROT.0.weight']
    Wm, bm = sd[prefix + 'mlp_h_mean_ROT.0.weight'], sd[prefix + 'mlp_h_mean_ROT.0.bias']
    Ts, bs, Yls, Yrs, As, ligs, status = [], [], [], [], [], [], []
    lo = ro = 0
    for pi, (nl, nr) in enumerate(zip(raw['lig_counts'], raw['rec_counts'])):    # :521-600
        H_r, H_l = h_r[ro:ro + nr], h_l[lo:lo + nl]
        Z_r, Z_l = x_r[ro:ro + nr], x_l[lo:lo + nl]
        q_r = _lrelu(_lin(H_r, Wm, bm), slope, f'{prefix}mlp_h_mean_ROT.r.{pi}').mean(0, keepdim=True)   # :524-525
        q_l = _lrelu(_lin(H_l, Wm, bm), slope, f'{prefix}mlp_h_mean_ROT.l.{pi}').mean(0, keepdim=True)   # :528-529
        att_r = torch.softmax(
            F.linear(H_r, Wk).view(-1, K, d).transpose(0, 1) @
            F.linear(q_l, Wq).view(1, K, d).transpose(0, 1).transpose(1, 2) / math.sqrt(d),
            dim=1).view(K, -1)                                               # :542-546
        Y_r = att_r @ Z_r                                                    # :548
        att_l = torch.softmax(
            F.linear(H_l, Wk).view(-1, K, d).transpose(0, 1) @
            F.linear(q_r, Wq).view(1, K, d).transpose(0, 1).transpose(1, 2) / math.sqrt(d),
            dim=1).view(K, -1)                                               # :553-557
        Y_l = att_l @ Z_l                                                    # :559
        T, b, A = kabsch(Y_r, Y_l, rand_fn, status)
        Ts.append(T); bs.append(b); Yls.append(Y_l); Yrs.append(Y_r); As.append(A)
        ligs.append((T @ raw['lig_x'][lo:lo + nl].t()).t() + b)              # :665
        lo += nl
        ro += nr
    return ligs, Yls, Yrs, Ts, bs, As, status


def forward(sd, args, raw, faithful=True, prefix='iegmn_original.', rand_fn=None, return_inter=False):
    """Rigid_Body_Docking_Net.forward (:642-692) for the single-stage (fine_tune=False) model.

    sd : state_dict with the reference's key names; raw: dict with lig_counts, rec_counts, and per
    side x / res / mu / src / dst / he tensors (ligand x is `new_x`).
    Returns (ligand_coords_list, keypts_ligand_list, keypts_receptor_list, rotation_list,
    translation_list) [+ intermediates]."""
    assert not args['fine_tune'] and args['dropout'] == 0.0 and args['nonlin'] == 'lkyrelu'
    slope = args['leakyrelu_neg_slope']
    L = args['iegmn_n_lays']
    x_l = x0_l = raw['lig_x']                                                # :452-456
    x_r = x0_r = raw['rec_x']
    emb = sd[prefix + 'residue_emb_layer.weight']
    h_l = emb[raw['lig_res'].view(-1).long()]                                # :459-462
    h_r = emb[raw['rec_res'].view(-1).long()]
    if args['use_mean_node_features']:
        h_l = torch.cat([h_l, torch.log(raw['lig_mu'])], dim=1)              # :467-471
        h_r = torch.cat([h_r, torch.log(raw['rec_mu'])], dim=1)
    h0_l, h0_r = h_l, h_r
    flag = 1.0 if args['use_edge_features_in_gmn'] else 0.0
    he_l = raw['ll_he'] * flag                                               # :480-481
    he_r = raw['rr_he'] * flag
    inter = {'layers': []}
    for i in range(L):                                                       # :483-501
        pfx = f'{prefix}iegmn_layers.{i}.'
        d_in = h_l.shape[1]
        li = {} if return_inter else None
        x_l, h_l, x_r, h_r = iegmn_layer(sd, pfx, args, d_in, raw, x_l, h_l, h0_l, he_l, x0_l,
                                         x_r, h_r, h0_r, he_r, x0_r, faithful, li)
        if return_inter:
            li.update(x_l=x_l, h_l=h_l, x_r=x_r, h_r=h_r)
            inter['layers'].append(li)

    ligs, Yls, Yrs, Ts, bs, As, status = head(sd, args, raw, h_l, x_l, h_r, x_r, prefix, rand_fn)
    outs = (ligs, Yls, Yrs, Ts, bs)
    if return_inter:
        inter.update(A=As, svd_iters=status)
        return outs, inter
    return outs


def scalar_loss(outs):
    """Fixed scalar loss of SURVEY.md section 8c: sum over pairs of mean(lig'^2)+mean(Yl^2)+mean(Yr^2)."""
    ligs, Yls, Yrs, _, _ = outs
    loss = 0.
    for lig, yl, yr in zip(ligs, Yls, Yrs):
        loss = loss + (lig ** 2).mean() + (yl ** 2).mean() + (yr ** 2).mean()
    return loss


def raw_from_graph(g):
    """PairGraph-like object (DGL accessor subset) -> the raw dict this port consumes."""
    return dict(
        lig_counts=[int(v) for v in g.batch_num_nodes('ligand')],
        rec_counts=[int(v) for v in g.batch_num_nodes('receptor')],
        lig_x=g.nodes['ligand'].data['new_x'].detach().cpu().float(),
        rec_x=g.nodes['receptor'].data['x'].detach().cpu().float(),
        lig_res=g.nodes['ligand'].data['res_feat'].detach().cpu(),
        rec_res=g.nodes['receptor'].data['res_feat'].detach().cpu(),
        lig_mu=g.nodes['ligand'].data['mu_r_norm'].detach().cpu().float(),
        rec_mu=g.nodes['receptor'].data['mu_r_norm'].detach().cpu().float(),
        ll_src=g.edge_endpoints('ll')[0].detach().cpu(), ll_dst=g.edge_endpoints('ll')[1].detach().cpu(),
        rr_src=g.edge_endpoints('rr')[0].detach().cpu(), rr_dst=g.edge_endpoints('rr')[1].detach().cpu(),
        ll_he=g.edges['ll'].data['he'].detach().cpu().float(),
        rr_he=g.edges['rr'].data['he'].detach().cpu().float(),
    )


def init_state_dict(args, seed, rot_scale=40.0):
    """PyTorch-default-initialised parameters with the reference's names and shapes
    (IEGMN_Layer.__init__ :82-175, IEGMN.__init__ :360-440), drawn in the reference's module
    construction order so that, for the same torch seed, they equal the reference's own init;
    att_mlp_{key,query}_ROT are then scaled x40 (keeps the SVD guard silent)."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    try:
        sd = {}
        d0 = args['residue_emb_dim'] + (5 if args['use_mean_node_features'] else 0)
        hid = args['iegmn_lay_hid_dim']
        emb = nn.Embedding(21, args['residue_emb_dim'])
        sd['iegmn_original.residue_emb_layer.weight'] = emb.weight.detach().clone()

        def layer(idx, d_in):
            p = f'iegmn_original.iegmn_layers.{idx}.'
            mods = [('edge_mlp.0', nn.Linear(2 * d_in + args['input_edge_feats_dim'] + 15, hid)),
                    ('edge_mlp.3', nn.LayerNorm(hid)), ('edge_mlp.4', nn.Linear(hid, hid)),
                    ('att_mlp_Q.0', nn.Linear(d_in, d_in, bias=False)),
                    ('att_mlp_K.0', nn.Linear(d_in, d_in, bias=False)),
                    ('att_mlp_V.0', nn.Linear(d_in, d_in, bias=False)),
                    ('node_mlp.0', nn.Linear(d0 + 2 * d_in + hid, d_in)), ('node_mlp.3', nn.LayerNorm(d_in)),
                    ('node_mlp.4', nn.Linear(d_in, hid)),
                    ('coors_mlp.0', nn.Linear(hid, hid)), ('coors_mlp.4', nn.Linear(hid, 1))]
            out = {}
            for name, m in mods:
                for k, v in m.state_dict().items():
                    out[p + name + '.' + k] = v.detach().clone()
            return out
        sd.update(layer(0, d0))
        L = args['iegmn_n_lays']
        if args['shared_layers']:
            shared = layer(1, hid)
            sd.update(shared)
            for i in range(2, L):
                for k, v in shared.items():
                    sd[k.replace('iegmn_layers.1.', f'iegmn_layers.{i}.')] = v
        else:
            for i in range(1, L):
                sd.update(layer(i, hid))
        K = args['num_att_heads']
        sd['iegmn_original.att_mlp_key_ROT.0.weight'] = nn.Linear(hid, K * hid, bias=False).weight.detach() * rot_scale
        sd['iegmn_original.att_mlp_query_ROT.0.weight'] = nn.Linear(hid, K * hid, bias=False).weight.detach() * rot_scale
        m = nn.Linear(hid, hid)
        sd['iegmn_original.mlp_h_mean_ROT.0.weight'] = m.weight.detach().clone()
        sd['iegmn_original.mlp_h_mean_ROT.0.bias'] = m.bias.detach().clone()
    finally:
        torch.set_rng_state(state)
    del g
    return sd
