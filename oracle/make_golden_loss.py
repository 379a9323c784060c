"""ORACLE-ONLY TOOL: tests/golden/loss_case.npz from the REAL reference functions, in THIS container.

src/train.py cannot be imported (import-time side effects: log directories, tensorboard, POT), so the three functions are
taken out of the reference's own source files with `ast` and executed as they are (nothing is copied into this
repository: only numbers leave this script): G_fn and compute_body_intersection_loss (src/train.py:41-49) and
compute_sq_dist_mat (src/utils/ot_utils.py:5-19).  Before writing, the restatement in oracle/loss_port.py must reproduce
every recorded number (<= 1e-6).

    python oracle/make_golden_loss.py
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import loss_port as port  # noqa: E402


def reference_functions(path, names):
    tree = ast.parse(open(path).read())
    ns = {'torch': torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return [ns[n] for n in names]


def main():
    g_fn, inter_fn = reference_functions('/root/reference/src/train.py', ['G_fn', 'compute_body_intersection_loss'])
    (sq_fn,) = reference_functions('/root/reference/src/utils/ot_utils.py', ['compute_sq_dist_mat'])
    mse_fn = torch.nn.MSELoss(reduction='mean')       # src/train.py:274
    rng = np.random.default_rng(123)
    sizes = [(40, 57), (1, 9), (130, 33), (64, 64)]
    sigma, ct = 25.0, 10.0                            # src/utils/args.py:69-70
    out = {'sizes': np.asarray(sizes), 'sigma': sigma, 'surface_ct': ct}
    for p, (nl, nr) in enumerate(sizes):
        # a ligand partly inside the receptor, so that both clamps are active for some points and inactive for others
        rec = rng.normal(0, 9.0, (nr, 3)).astype(np.float32)
        tgt = (rng.normal(0, 7.0, (nl, 3)) + np.array([14.0, 0, 0])).astype(np.float32)
        pred = (tgt + rng.normal(0, 2.0, (nl, 3)) - np.array([6.0, 0, 0])).astype(np.float32)
        a = torch.tensor(pred, requires_grad=True)
        t, r = torch.tensor(tgt), torch.tensor(rec)
        mse, inter = mse_fn(a, t), inter_fn(a, r, sigma, ct)
        (gm,) = torch.autograd.grad(mse, a, retain_graph=True)
        (gi,) = torch.autograd.grad(inter, a)
        kp = torch.tensor(rng.normal(0, 5.0, (50, 3)).astype(np.float32))
        sq = sq_fn(t[: min(nl, 20)], kp)
        # the restatement must agree with the reference's functions
        a2 = torch.tensor(pred, requires_grad=True)
        mse2, inter2 = port.mse_loss(a2, t), port.body_intersection_loss(a2, r, sigma, ct)
        (gm2,) = torch.autograd.grad(mse2, a2, retain_graph=True)
        (gi2,) = torch.autograd.grad(inter2, a2)
        for x, y, what in ((mse, mse2, 'mse'), (inter, inter2, 'inter'), (gm, gm2, 'dmse'), (gi, gi2, 'dinter'),
                           (sq, port.sq_dist_mat(t[: min(nl, 20)], kp), 'sq'), (g_fn(r, a, sigma), port.g_fn(r, a2, sigma), 'G')):
            assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(x.abs().max())), (p, what)
        out.update({f'pred{p}': pred, f'tgt{p}': tgt, f'rec{p}': rec, f'mse{p}': mse.detach().numpy(),
                    f'inter{p}': inter.detach().numpy(), f'dmse{p}': gm.numpy(), f'dinter{p}': gi.numpy(),
                    f'kp{p}': kp.numpy(), f'sq{p}': sq.numpy()})
        print(p, (nl, nr), 'mse', float(mse), 'inter', float(inter))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'loss_case.npz'), **out)


if __name__ == '__main__':
    main()
