"""The oracle (oracle/iegmn_port.py) against the golden vectors captured from the imported
reference (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import iegmn_port as port
from tests.util import CASES, REAL_CASES, cat_out, load_case, state_dict_for


@pytest.mark.parametrize('name', CASES + REAL_CASES)
@pytest.mark.parametrize('faithful', [True, False])
def test_port_forward_matches_reference(name, faithful):
    z, meta, args, raw = load_case(name)
    sd = state_dict_for(meta, args)
    rand_fn = None
    if 'svd_draws' in z.files:
        draws = [torch.from_numpy(d) for d in z['svd_draws']]
        rand_fn = lambda n: draws[n]  # noqa: E731
    with torch.no_grad():
        outs, inter = port.forward(sd, args, raw, faithful=faithful, rand_fn=rand_fn, return_inter=True)
    assert inter['svd_iters'] == meta['svd_iters']
    for nm, lst in zip(('lig', 'Yl', 'Yr', 'T', 'b'), outs):
        got = cat_out(lst).numpy()
        # (real structures: coordinates up to 70 A, and a batch-wide softmax row of the reference sums in another order than the
        #  per-pair rows of the block-diagonal form - 6e-6 of the output's magnitude on the two-complex batch)
        atol = 1e-5 * max(1.0, float(np.abs(z['out_' + nm]).max())) if name in REAL_CASES else 1e-5
        np.testing.assert_allclose(got, z['out_' + nm], rtol=1e-5, atol=atol, err_msg=nm)
    L = args['iegmn_n_lays']
    for i in sorted({0, 1, L - 1}):
        li = inter['layers'][i]
        for key, got in ((f'layer{i}_x', torch.cat([li['x_l'], li['x_r']])), (f'layer{i}_h', torch.cat([li['h_l'], li['h_r']]))):
            atol = 1e-5 * max(1.0, float(np.abs(z[key]).max())) if name in REAL_CASES else 1e-5
            np.testing.assert_allclose(got.numpy(), z[key], rtol=1e-5, atol=atol, err_msg=key)


@pytest.mark.parametrize('name', ['A_b1_shared5', 'D_degraded3', 'F_real_1GL1'])
def test_port_gradients_match_reference(name):
    z, meta, args, raw = load_case(name)
    sd = state_dict_for(meta, args)
    leaves = {}
    sdp = {}
    for k, v in sd.items():
        key = k
        if args['shared_layers'] and '.iegmn_layers.' in k and int(k.split('.')[2]) >= 2:
            key = k.replace(f".iegmn_layers.{k.split('.')[2]}.", '.iegmn_layers.1.')
        if key not in leaves:
            leaves[key] = v.clone().requires_grad_(True)
        sdp[k] = leaves[key]
    outs = port.forward(sdp, args, raw, faithful=True)
    loss = port.scalar_loss(outs)
    loss.backward()
    assert abs(float(loss.detach()) - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    for k in z.files:
        if not k.startswith('grad_'):
            continue
        g = leaves[k[5:]].grad.numpy()
        ref = z[k]
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(g - ref).max() <= 2e-4 * scale, k


def test_reference_invariants():
    """Runtime self-checks of the reference: T T^T = I, det T = +1, b is (1,3)."""
    z, meta, args, raw = load_case('B_b3_dips8')
    T = z['out_T'].reshape(-1, 3, 3)
    for t in T:
        np.testing.assert_allclose(t @ t.T, np.eye(3), atol=1e-5)
        assert abs(np.linalg.det(t) - 1.0) < 1e-5
    assert z['out_b'].shape == (len(raw['lig_counts']), 3)


def test_loss_port_matches_reference_vectors():
    """oracle/loss_port.py against the numbers recorded from the reference's own G_fn / compute_body_intersection_loss /
    compute_sq_dist_mat / nn.MSELoss (tests/golden/loss_case.npz, oracle/make_golden_loss.py)"""
    import os
    import numpy as np
    import torch
    from oracle import loss_port as lp
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'loss_case.npz'))
    sigma, ct = float(z['sigma']), float(z['surface_ct'])
    for p in range(len(z['sizes'])):
        a = torch.tensor(z[f'pred{p}'], requires_grad=True)
        t, r = torch.tensor(z[f'tgt{p}']), torch.tensor(z[f'rec{p}'])
        mse, inter = lp.mse_loss(a, t), lp.body_intersection_loss(a, r, sigma, ct)
        (gm,) = torch.autograd.grad(mse, a, retain_graph=True)
        (gi,) = torch.autograd.grad(inter, a)
        assert abs(float(mse.detach()) - float(z[f'mse{p}'])) <= 1e-6 * max(1.0, abs(float(mse.detach())))
        assert abs(float(inter.detach()) - float(z[f'inter{p}'])) <= 1e-6 * max(1.0, abs(float(inter.detach())))
        assert np.allclose(gm.numpy(), z[f'dmse{p}'], atol=1e-6) and np.allclose(gi.numpy(), z[f'dinter{p}'], atol=1e-6)
        nl = a.shape[0]
        assert np.allclose(lp.sq_dist_mat(t[: min(nl, 20)], torch.tensor(z[f'kp{p}'])).numpy(), z[f'sq{p}'], rtol=1e-6, atol=1e-5)
