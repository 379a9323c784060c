"""ORACLE (test infrastructure): the pocket optimal-transport term of the reference's loss, restated.

Only tests/ may import this file; the product (equidock_public_amd/losses.py) never does.

Reference: src/utils/ot_utils.py:5-29 (`compute_sq_dist_mat`, `compute_ot_emd`) called per pair at src/train.py:117-129:
    cost = sq_dist(pocket_lig, Y_lig) + sq_dist(pocket_rec, Y_rec)            (n_pocket x K)
    plan = ot.emd(1/n, 1/K, cost.detach().cpu().numpy(), numItermax=10000)     # POT, exact network simplex, on the host
    ot_dist = sum(plan.float() * cost)                                          # gradient flows through `cost` only

PARITY UNPINNED for the solver: `ot` is POT==0.7.0 (requirements.txt), a third-party C++ network simplex that is neither
under /root/reference nor installed here.  Its published contract is "an exact solution of the earth mover's distance
LP"; this oracle therefore solves the same linear program
    min <P, M>  s.t.  P 1 = a,  P^T 1 = b,  P >= 0
with scipy's HiGHS (dual simplex, an independent exact LP solver).  For cost matrices in general position the optimal
plan is unique, so any exact solver - POT's, HiGHS, the product's successive-shortest-path solver - returns the same
plan; the optimal VALUE is unique always.  `compute_sq_dist_mat` is pinned: tests/golden/loss_case.npz holds vectors
recorded from the reference's own function (oracle/make_golden_loss.py).
"""
import numpy as np
import torch


def compute_sq_dist_mat(X_1, X_2):
    """src/utils/ot_utils.py:5-19."""
    return ((X_1.view(X_1.shape[0], 1, -1) - X_2.view(1, X_2.shape[0], -1)) ** 2).sum(dim=2)


def emd_lp(a, b, M):
    """Exact optimal plan of the transportation LP (HiGHS dual simplex). a [n], b [m], M [n, m] -> plan [n, m] float64."""
    from scipy.optimize import linprog
    from scipy.sparse import lil_matrix
    n, m = M.shape
    A = lil_matrix((n + m, n * m))
    for i in range(n):
        A[i, i * m:(i + 1) * m] = 1.0
    for j in range(m):
        A[n + j, j::m] = 1.0
    res = linprog(np.asarray(M, dtype=np.float64).reshape(-1), A_eq=A.tocsr(), b_eq=np.concatenate([a, b]),
                  bounds=(0, None), method='highs-ds')
    assert res.status == 0, res.message
    return res.x.reshape(n, m)


def compute_ot_emd(cost_mat):
    """src/utils/ot_utils.py:22-29 with the LP solver in place of ot.emd. Returns (ot_dist, plan float32 tensor)."""
    n, m = cost_mat.shape
    plan = emd_lp(np.ones(n) / n, np.ones(m) / m, cost_mat.detach().cpu().numpy())
    plan_t = torch.tensor(plan).float()
    return torch.sum(plan_t * cost_mat), plan_t


def pocket_ot_loss(pocket_lig, pocket_rec, Y_lig, Y_rec):
    """src/train.py:117-129 for one pair: (N, 3) pocket coordinates of both proteins, (K, 3) keypoints."""
    cost = compute_sq_dist_mat(pocket_lig, Y_lig) + compute_sq_dist_mat(pocket_rec, Y_rec)
    return compute_ot_emd(cost)
