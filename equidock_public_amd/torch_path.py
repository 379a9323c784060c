"""The reference's NON-published model options through torch operators on the tensors' device.

The published configuration family runs as two C calls into the HIP library (model.py: _IEGMNFunction).  The reference's
other switches - nonlin 'swish', layer_norm 'BN' / '0', layer_norm_coors 'LN' / 'BN', final_h_layer_norm 'BN' / 'LN' / 'GN',
the fine-tune stage (rigid_docking_model.py:161-174, 294-310, 622-627) and dropout > 0 in training mode (its masks come
from torch's generator, which a fused kernel cannot reproduce) - are rare side branches ("didn't work",
src/utils/args.py:110), so they are not worth kernels of their own; but a user of the reference must still be able to
construct, load and train such a model.  This module evaluates the SAME parameter-holding sub-modules the constructors in
model.py build (nn.Sequential with the reference's Dropout / norm / non-linearity members, hence its RNG consumption
order and BatchNorm statistics) with plain torch operators, without DGL:

  * u_sub_v / gathers        -> index selects on the batch's int32 endpoints
  * copy_edge + mean         -> index_add + degree clamp (zero for in-degree 0)
  * dense masked attention   -> per-pair softmax (equal unless every in-pair logit is < -900)
  * per-pair keypoints + SVD -> the reference's loop, incl. its guard loop drawing torch.rand(3, 3)

It is NOT a fallback of the HIP path: which path runs is decided by the configuration alone (model.hip_path_supported),
never by the availability of the library.  Pinned against vectors recorded from the reference itself
(tests/golden/variants.npz, oracle/make_golden_variants.py).
"""
import math

import torch

RBF_SIGMAS = [1.5 ** k for k in range(15)]      # rigid_docking_model.py:116


def _mean_by_dst(val, dst, n):
    acc = torch.zeros((n,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device).index_add(0, dst, val)
    deg = torch.zeros(n, dtype=val.dtype, device=val.device).index_add(
        0, dst, torch.ones(dst.numel(), dtype=val.dtype, device=val.device))
    return acc / deg.clamp(min=1.0).view(n, *([1] * (val.dim() - 1)))


def _cross_attention(queries, keys, values, q_counts, k_counts, cross_msgs):
    """compute_cross_attention (rigid_docking_model.py:46-64), block-diagonal."""
    if not cross_msgs:
        return queries * 0.
    outs, qo, ko = [], 0, 0
    for nq, nk in zip(q_counts, k_counts):
        a = torch.mm(queries[qo:qo + nq], keys[ko:ko + nk].t())
        outs.append(torch.mm(torch.softmax(a, dim=1), values[ko:ko + nk]))
        qo += nq
        ko += nk
    return torch.cat(outs, 0)


def _final_norm(layer, counts, h):
    if layer.final_h_layer_norm == 'GN':
        return layer.final_h_layernorm_layer(counts, h)
    return layer.final_h_layernorm_layer(h)


def layer_forward(layer, g, x_l, h_l, h0_l, he_l, x0_l, x_r, h_r, h0_r, he_r, x0_r):
    """IEGMN_Layer.forward (rigid_docking_model.py:189-352) -> x_ligand', h_ligand', x_receptor', h_receptor'.
    `g`: PairGraph batch (endpoints + per-pair counts); sub-modules are called in the reference's order."""
    lc, rc = g._batch_nodes['ligand'], g._batch_nodes['receptor']
    sides = {}
    for side, et, x, h, he in (('l', 'll', x_l, h_l, he_l), ('r', 'rr', x_r, h_r, he_r)):
        src, dst = (t.long() for t in g._edges[et])
        x_rel = x[src] - x[dst]                                                  # :204-205
        d2 = torch.sum(x_rel ** 2, dim=1, keepdim=True)                          # :208-209
        rbf = torch.cat([torch.exp(-d2 / s) for s in RBF_SIGMAS], dim=-1)        # :210-214
        if not layer.use_dist_in_layers:
            rbf = rbf * 0.                                                       # :216-218
        sides[side] = dict(src=src, dst=dst, x_rel=x_rel, cat=torch.cat([h[src], h[dst], he, rbf], dim=-1), n=x.shape[0])
    for side in ('l', 'r'):                                                      # :236-237 (ligand edges first)
        sides[side]['msg'] = layer.edge_mlp(sides[side]['cat'])
    cross_l = _cross_attention(layer.att_mlp_Q(h_l), layer.att_mlp_K(h_r), layer.att_mlp_V(h_r), lc, rc, layer.cross_msgs)
    cross_r = _cross_attention(layer.att_mlp_Q(h_r), layer.att_mlp_K(h_l), layer.att_mlp_V(h_l), rc, lc, layer.cross_msgs)
    for side in ('l', 'r'):                                                      # :263-266
        s = sides[side]
        s['coef'] = layer.coors_mlp(s['msg'])
    for side in ('l', 'r'):                                                      # :274-283
        s = sides[side]
        s['x_update'] = _mean_by_dst(s['x_rel'] * s['coef'], s['dst'], s['n'])
        s['aggr_msg'] = _mean_by_dst(s['msg'], s['dst'], s['n'])
    eta = layer.x_connection_init
    xf_l = eta * x0_l + (1. - eta) * x_l + sides['l']['x_update']               # :286-292
    xf_r = eta * x0_r + (1. - eta) * x_r + sides['r']['x_update']
    if layer.fine_tune:                                                          # :294-310
        xf_l = xf_l + layer.att_mlp_cross_coors_V(h_l) * (
            x_l - _cross_attention(layer.att_mlp_cross_coors_Q(h_l), layer.att_mlp_cross_coors_K(h_r), x_r, lc, rc,
                                   layer.cross_msgs))
        xf_r = xf_r + layer.att_mlp_cross_coors_V(h_r) * (
            x_r - _cross_attention(layer.att_mlp_cross_coors_Q(h_r), layer.att_mlp_cross_coors_K(h_l), x_l, rc, lc,
                                   layer.cross_msgs))
    in_l = torch.cat((layer.node_norm(h_l), sides['l']['aggr_msg'], cross_l, h0_l), dim=-1)     # :319-329
    in_r = torch.cat((layer.node_norm(h_r), sides['r']['aggr_msg'], cross_r, h0_r), dim=-1)
    if layer.h_feats_dim == layer.out_feats_dim:                                 # :332-337
        up_l = layer.skip_weight_h * layer.node_mlp(in_l) + (1. - layer.skip_weight_h) * h_l
        up_r = layer.skip_weight_h * layer.node_mlp(in_r) + (1. - layer.skip_weight_h) * h_r
    else:
        up_l = layer.node_mlp(in_l)
        up_r = layer.node_mlp(in_r)
    up_l = _final_norm(layer, lc, up_l)                                          # :348-349
    up_r = _final_norm(layer, rc, up_r)
    return xf_l, up_l, xf_r, up_r


def iegmn_forward(iegmn, g):
    """IEGMN.forward (rigid_docking_model.py:451-602) + the rigid apply of Rigid_Body_Docking_Net.forward (:665) for
    one stage.  Returns batched T [B, 3, 3], b [B, 3], Y_lig, Y_rec [B, K, 3], lig' [n_lig, 3]."""
    nl, nr = g._ndata['ligand'], g._ndata['receptor']
    x_l, x_r = nl['new_x'], nr['x']                                              # :452-456
    x0_l, x0_r = x_l, x_r
    h_l = iegmn.residue_emb_layer(nl['res_feat'].view(-1).long())                # :459-462
    h_r = iegmn.residue_emb_layer(nr['res_feat'].view(-1).long())
    if iegmn.use_mean_node_features:
        h_l = torch.cat([h_l, torch.log(nl['mu_r_norm'])], dim=1)                # :467-471
        h_r = torch.cat([h_r, torch.log(nr['mu_r_norm'])], dim=1)
    h0_l, h0_r = h_l, h_r
    flag = 1.0 if iegmn.use_edge_features_in_gmn else 0.0
    he_l = g._edata['ll']['he'] * flag                                           # :480-481
    he_r = g._edata['rr']['he'] * flag
    for layer in iegmn.iegmn_layers:                                             # :483-501
        x_l, h_l, x_r, h_r = layer_forward(layer, g, x_l, h_l, h0_l, he_l, x0_l, x_r, h_r, h0_r, he_r, x0_r)
    K, d = iegmn.num_att_heads, iegmn.out_feats_dim
    Ts, bs, Yls, Yrs, ligs = [], [], [], [], []
    lo = ro = 0
    for n_l, n_r in zip(g._batch_nodes['ligand'], g._batch_nodes['receptor']):     # :521-600
        H_r, H_l = h_r[ro:ro + n_r], h_l[lo:lo + n_l]
        Z_r, Z_l = x_r[ro:ro + n_r], x_l[lo:lo + n_l]
        q_r = torch.mean(iegmn.mlp_h_mean_ROT(H_r), dim=0, keepdim=True)          # :524-525 (receptor first)
        q_l = torch.mean(iegmn.mlp_h_mean_ROT(H_l), dim=0, keepdim=True)          # :528-529
        att_r = torch.softmax(iegmn.att_mlp_key_ROT(H_r).view(-1, K, d).transpose(0, 1) @
                              iegmn.att_mlp_query_ROT(q_l).view(1, K, d).transpose(0, 1).transpose(1, 2) / math.sqrt(d),
                              dim=1).view(K, -1)                                 # :542-546
        Y_r = att_r @ Z_r
        att_l = torch.softmax(iegmn.att_mlp_key_ROT(H_l).view(-1, K, d).transpose(0, 1) @
                              iegmn.att_mlp_query_ROT(q_r).view(1, K, d).transpose(0, 1).transpose(1, 2) / math.sqrt(d),
                              dim=1).view(K, -1)                                 # :553-557
        Y_l = att_l @ Z_l
        Yr_mean, Yl_mean = Y_r.mean(dim=0, keepdim=True), Y_l.mean(dim=0, keepdim=True)     # :563-564
        A = (Y_r - Yr_mean).t() @ (Y_l - Yl_mean)                                # :567
        if torch.isnan(A).any():
            raise RuntimeError("NaN in the Kabsch covariance (rigid_docking_model.py:570)")
        U, S, Vt = torch.linalg.svd(A)                                           # :571
        num_it = 0
        eye = torch.eye(3, device=A.device)
        while torch.min(S) < 1e-3 or torch.min(torch.abs((S ** 2).view(1, 3) - (S ** 2).view(3, 1) + eye)) < 1e-2:
            A = A + torch.rand(3, 3).to(A.device) * eye                          # :578
            U, S, Vt = torch.linalg.svd(A)
            num_it += 1
            if num_it > 10:
                raise RuntimeError("SVD consistently numerically unstable (rigid_docking_model.py:582-584)")
        corr = torch.diag(torch.tensor([1., 1., float(torch.sign(torch.det(A.detach())))], device=A.device))   # :586
        T = (U @ corr) @ Vt                                                      # :587
        b = Yr_mean - torch.t(T @ Yl_mean.t())                                   # :589
        Ts.append(T); bs.append(b.view(3)); Yls.append(Y_l); Yrs.append(Y_r)
        ligs.append((T @ x0_l[lo:lo + n_l].t()).t() + b)                         # :665
        lo += n_l
        ro += n_r
    return torch.stack(Ts), torch.stack(bs), torch.stack(Yls), torch.stack(Yrs), torch.cat(ligs, 0)
