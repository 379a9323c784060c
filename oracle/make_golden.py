"""ORACLE-ONLY TOOL: generate tests/golden/*.npz from the REAL reference, in THIS container.

Runs /root/reference/src/model/rigid_docking_model.py *unmodified* on CPU (with the DGL
stand-in of oracle/_dgl_standin first on sys.path) on seeded synthetic pair graphs, records
inputs, outputs, per-layer states and parameter gradients of the fixed scalar loss
(SURVEY.md section 8c), and -- before writing anything -- asserts that the restatement in
oracle/iegmn_port.py reproduces every recorded tensor (<= 1e-5).  The reference sources are
never copied: only numbers leave this script.

    python oracle/make_golden.py            # writes tests/golden/case_*.npz

/root/reference does not exist on the GPU box; nothing at test/bench time needs this script.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, '_dgl_standin'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)

import dgl  # noqa: E402  (the stand-in)
from src.model import rigid_docking_model as ref  # noqa: E402  (the real reference)

from equidock_public_amd import synthetic  # noqa: E402
from oracle import iegmn_port as port  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def ref_graph(pairs):
    """hetero_graph_from_sg_l_r_pair + dgl.batch (src/utils/train_utils.py:61-100) on the stand-in."""
    gs = []
    for lig, rec in pairs:
        g = dgl.heterograph({
            ('ligand', 'll', 'ligand'): (torch.from_numpy(lig['src']), torch.from_numpy(lig['dst'])),
            ('receptor', 'rr', 'receptor'): (torch.from_numpy(rec['src']), torch.from_numpy(rec['dst'])),
            ('receptor', 'cross', 'ligand'): (torch.tensor([], dtype=torch.int32), torch.tensor([], dtype=torch.int32)),
            ('ligand', 'cross', 'receptor'): (torch.tensor([], dtype=torch.int32), torch.tensor([], dtype=torch.int32)),
        }, num_nodes_dict={'ligand': lig['x'].shape[0], 'receptor': rec['x'].shape[0]})
        for k in ('res_feat', 'x', 'new_x', 'mu_r_norm'):
            g.nodes['ligand'].data[k] = torch.from_numpy(lig[k])
        for k in ('res_feat', 'x', 'mu_r_norm'):
            g.nodes['receptor'].data[k] = torch.from_numpy(rec[k])
        g.edges['ll'].data['he'] = torch.from_numpy(lig['he'])
        g.edges['rr'].data['he'] = torch.from_numpy(rec['he'])
        gs.append(g)
    return dgl.batch(gs)


def raw_from_pairs(pairs):
    cat = lambda side, k: torch.from_numpy(np.concatenate([p[side][k] for p in pairs], 0))  # noqa: E731
    def edges(side, k, nodes_key='x'):
        off, out = 0, []
        for p in pairs:
            out.append(p[side][k].astype(np.int64) + off)
            off += p[side][nodes_key].shape[0]
        return torch.from_numpy(np.concatenate(out))
    return dict(
        lig_counts=[p[0]['x'].shape[0] for p in pairs], rec_counts=[p[1]['x'].shape[0] for p in pairs],
        lig_x=cat(0, 'new_x'), rec_x=cat(1, 'x'), lig_res=cat(0, 'res_feat'), rec_res=cat(1, 'res_feat'),
        lig_mu=cat(0, 'mu_r_norm'), rec_mu=cat(1, 'mu_r_norm'),
        ll_src=edges(0, 'src'), ll_dst=edges(0, 'dst'), rr_src=edges(1, 'src'), rr_dst=edges(1, 'dst'),
        ll_he=cat(0, 'he'), rr_he=cat(1, 'he'))


def fingerprint(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def run_case(name, sizes, seed, args_over, rot_scale=40.0, degrade=False, full_grads=True, record_draws=False, pairs=None,
             extra=None, note=None):
    """pairs: given per-pair (ligand, receptor) arrays instead of the synthetic generator's (oracle/make_golden_real.py: the
    graphs the reference's OWN builder made from real structures); extra: more arrays for the fixture."""
    args = port.default_args(**args_over)
    if pairs is None:
        pairs = synthetic.make_pairs(sizes, seed)
    if degrade:
        # in-degree < 10 for some nodes and one isolated node (zero-fill semantics of fn.mean)
        for lig, rec in pairs:
            for p in (lig, rec):
                keep = np.ones(len(p['dst']), dtype=bool)
                keep[p['dst'] == 3] = False                     # node 3: isolated as a destination
                keep[(p['dst'] == 5) & (np.arange(len(keep)) % 2 == 0)] = False
                keep[(p['dst'] == 7) & (np.arange(len(keep)) % 3 != 0)] = False
                for k in ('src', 'dst', 'he'):
                    p[k] = p[k][keep]
    torch.manual_seed(seed)
    model = ref.Rigid_Body_Docking_Net(args=args, log=print)
    with torch.no_grad():
        model.iegmn_original.att_mlp_key_ROT[0].weight.mul_(rot_scale)
        model.iegmn_original.att_mlp_query_ROT[0].weight.mul_(rot_scale)
    sd_ref = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd = port.init_state_dict(args, seed, rot_scale)
    assert set(sd) == set(sd_ref), (set(sd) ^ set(sd_ref))
    for k in sd:
        assert torch.equal(sd[k], sd_ref[k]), f'init mismatch {k}'

    # record per-layer outputs and the cross-attention outputs of the reference
    layer_out, cross_out = [], []
    hooks = []
    seen = set()
    for lay in model.iegmn_original.iegmn_layers:
        if id(lay) in seen:
            continue
        seen.add(id(lay))
        hooks.append(lay.register_forward_hook(lambda m, i, o: layer_out.append([t.detach().clone() for t in o])))
    orig_cca = ref.compute_cross_attention

    def rec_cca(*a, **k):
        out = orig_cca(*a, **k)
        cross_out.append(out.detach().clone())
        return out
    ref.compute_cross_attention = rec_cca
    draws = []
    orig_rand = torch.rand
    if record_draws:
        def rec_rand(*a, **k):
            out = orig_rand(*a, **k)
            if tuple(out.shape) == (3, 3):
                draws.append(out.clone())
            return out
        torch.rand = rec_rand
    try:
        g = ref_graph(pairs)
        model.zero_grad()
        outs = model(g, epoch=0)
        loss = port.scalar_loss(outs)
        loss.backward()
    finally:
        ref.compute_cross_attention = orig_cca
        torch.rand = orig_rand
        for h in hooks:
            h.remove()
    grads_ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    # ---- check the restatement against the reference before writing anything ----------------
    raw = raw_from_pairs(pairs)
    sdp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    if args['shared_layers']:      # shared tensors must be the same leaf
        for k in list(sdp):
            if '.iegmn_layers.' in k and int(k.split('.')[2]) >= 2:
                sdp[k] = sdp[k.replace(f".iegmn_layers.{k.split('.')[2]}.", '.iegmn_layers.1.')]
    it = iter(draws)
    outs_p, inter = port.forward(sdp, args, raw, faithful=True, return_inter=True,
                                 rand_fn=(lambda n: next(it)) if record_draws else None)
    port.scalar_loss(outs_p).backward()
    worst = 0.0
    for a, b in zip(outs, outs_p):
        for x, y in zip(a, b):
            worst = max(worst, float((x - y).abs().max()))
    L = args['iegmn_n_lays']
    for i in range(L):
        xl, hl, xr, hr = layer_out[i]
        li = inter['layers'][i]
        worst = max(worst, float((xl - li['x_l']).abs().max()), float((hl - li['h_l']).abs().max()),
                    float((xr - li['x_r']).abs().max()), float((hr - li['h_r']).abs().max()))
    gworst = 0.0
    for k, gr in grads_ref.items():
        gp = sdp[k].grad
        rel = float((gr - gp).abs().max()) / (float(gr.abs().max()) + 1e-12)
        gworst = max(gworst, rel)
    print(f'[{name}] port vs reference: outputs/layers max abs diff {worst:.3e}, grads max rel diff {gworst:.3e}, '
          f'svd perturbations {inter["svd_iters"]}')
    assert worst < 2e-4 and gworst < 2e-3, (worst, gworst)
    # block-diagonal variant must agree too (what the HIP path computes)
    outs_b = port.forward(sd, args, raw, faithful=False,
                          rand_fn=(lambda n, it2=iter(draws): next(it2)) if record_draws else None)
    wb = max(float((x - y).abs().max()) for a, b in zip(outs, outs_b) for x, y in zip(a, b))
    # relative to each output's largest magnitude (coordinates of real structures reach 70 A; the batch-wide softmax rows of the
    # reference sum in another order than per-pair rows, fp32 rounding that 8 layers and the keypoint softmax carry to the outputs)
    wbr = max(float((x - y).abs().max()) / max(1.0, float(x.abs().max())) for a, b in zip(outs, outs_b) for x, y in zip(a, b))
    print(f'[{name}] block-diagonal attention vs reference: max abs diff {wb:.3e}, relative to the output\'s largest magnitude {wbr:.3e}')
    assert wbr < 2e-5, (wb, wbr)

    # ---- write the fixture ------------------------------------------------------------------
    blob = {}
    meta = dict(name=name, sizes=[list(s) for s in sizes], seed=seed, rot_scale=rot_scale,
                args={k: v for k, v in args.items() if k != 'device'}, fingerprint=fingerprint(sd_ref),
                svd_iters=inter['svd_iters'], torch=torch.__version__, degrade=degrade,
                note=note or 'generated by oracle/make_golden.py from the imported reference + DGL stand-in')
    for k, v in raw.items():
        if torch.is_tensor(v):
            blob['in_' + k] = v.numpy()
    blob['in_lig_counts'] = np.asarray(raw['lig_counts'], np.int32)
    blob['in_rec_counts'] = np.asarray(raw['rec_counts'], np.int32)
    names = ('lig', 'Yl', 'Yr', 'T', 'b')
    for nm, lst in zip(names, outs):
        blob['out_' + nm] = torch.cat([t.detach().reshape(-1, t.shape[-1]) for t in lst], 0).numpy()
    for i in sorted({0, 1, L - 1}):
        xl, hl, xr, hr = layer_out[i]
        blob[f'layer{i}_x'] = torch.cat([xl, xr], 0).numpy()
        blob[f'layer{i}_h'] = torch.cat([hl, hr], 0).numpy()
    if len(cross_out) >= 4:
        blob['layer1_cross'] = torch.cat([cross_out[2], cross_out[3]], 0).numpy()
    blob['loss'] = np.asarray(float(loss))
    if record_draws:
        blob['svd_draws'] = torch.stack(draws).numpy() if draws else np.zeros((0, 3, 3), np.float32)
    gnorm = {}
    for k, gr in grads_ref.items():
        gnorm[k] = [float(gr.double().sum()), float(gr.double().norm())]
        if full_grads:
            blob['grad_' + k] = gr.numpy()
    meta['grad_fingerprint'] = gnorm
    blob['meta'] = np.asarray(json.dumps(meta))
    blob.update(extra or {})
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, f'case_{name}.npz')
    np.savez_compressed(path, **blob)
    print(f'[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


def main():
    # 5-layer shared / skH 0.5 = the DB5.5 checkpoint's configuration (src/inference_rigid.py:93)
    run_case('A_b1_shared5', [(57, 83)], 0, dict(iegmn_n_lays=5, shared_layers=True, skip_weight_h=0.5))
    # 8-layer non-shared / skH 0.75 = the DIPS checkpoint's configuration (src/inference_rigid.py:90)
    run_case('B_b3_dips8', [(40, 121), (220, 64), (97, 150)], 1, dict(iegmn_n_lays=8, skip_weight_h=0.75))
    run_case('C_b2_200', [(200, 200), (200, 200)], 2, dict(iegmn_n_lays=8, skip_weight_h=0.75), full_grads=False)
    run_case('D_degraded3', [(33, 47), (52, 29)], 3, dict(iegmn_n_lays=3, skip_weight_h=0.5), degrade=True)
    # default-scale ROT weights: keypoints collapse, the SVD guard (:574-580) fires; draws recorded
    run_case('E_svd_guard', [(45, 60)], 4, dict(iegmn_n_lays=2, skip_weight_h=0.5), rot_scale=1.0,
             record_draws=True, full_grads=False)


if __name__ == '__main__':
    main()
