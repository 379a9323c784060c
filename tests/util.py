"""Shared helpers for the tests: golden-fixture loading and comparisons."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

CASES = ['A_b1_shared5', 'B_b3_dips8', 'C_b2_200', 'D_degraded3', 'E_svd_guard']
# the real reference model on graphs the reference's own builder made from real DB5.5 structures (oracle/make_golden_real.py)
REAL_CASES = ['F_real_1GL1', 'F_real_2J7P', 'F_real_batch2']


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f'case_{name}.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    args = dict(meta['args'])
    args['device'] = torch.device('cpu')
    raw = {}
    for k in z.files:
        if k.startswith('in_'):
            raw[k[3:]] = torch.from_numpy(z[k])
    raw['lig_counts'] = [int(v) for v in z['in_lig_counts']]
    raw['rec_counts'] = [int(v) for v in z['in_rec_counts']]
    return z, meta, args, raw


def state_dict_for(meta, args):
    """Regenerate the fixture's parameters from its seed and check their fingerprint."""
    from oracle import iegmn_port as port
    sd = port.init_state_dict(args, meta['seed'], meta['rot_scale'])
    for k, (s, a) in meta['fingerprint'].items():
        v = sd[k].double()
        assert abs(float(v.sum()) - s) <= 1e-9 * max(1.0, abs(a)), f'parameter fingerprint mismatch: {k}'
        assert abs(float(v.abs().sum()) - a) <= 1e-9 * max(1.0, abs(a)), f'parameter fingerprint mismatch: {k}'
    return sd


def pairs_from_raw(raw):
    """Split the fixture's concatenated inputs back into per-pair (ligand, receptor) dicts."""
    pairs = []
    lo = ro = leo = reo = 0
    ll_dst, rr_dst = raw['ll_dst'].numpy(), raw['rr_dst'].numpy()
    for nl, nr in zip(raw['lig_counts'], raw['rec_counts']):
        le = int(((ll_dst >= lo) & (ll_dst < lo + nl)).sum())
        re = int(((rr_dst >= ro) & (rr_dst < ro + nr)).sum())
        lig = dict(x=raw['lig_x'][lo:lo + nl].numpy(), new_x=raw['lig_x'][lo:lo + nl].numpy(),
                   res_feat=raw['lig_res'][lo:lo + nl].numpy(), mu_r_norm=raw['lig_mu'][lo:lo + nl].numpy(),
                   src=(raw['ll_src'][leo:leo + le] - lo).numpy().astype(np.int32),
                   dst=(raw['ll_dst'][leo:leo + le] - lo).numpy().astype(np.int32),
                   he=raw['ll_he'][leo:leo + le].numpy())
        rec = dict(x=raw['rec_x'][ro:ro + nr].numpy(),
                   res_feat=raw['rec_res'][ro:ro + nr].numpy(), mu_r_norm=raw['rec_mu'][ro:ro + nr].numpy(),
                   src=(raw['rr_src'][reo:reo + re] - ro).numpy().astype(np.int32),
                   dst=(raw['rr_dst'][reo:reo + re] - ro).numpy().astype(np.int32),
                   he=raw['rr_he'][reo:reo + re].numpy())
        pairs.append((lig, rec))
        lo += nl; ro += nr; leo += le; reo += re
    return pairs


def cat_out(lst):
    return torch.cat([t.reshape(-1, t.shape[-1]) for t in lst], 0)
