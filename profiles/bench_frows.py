"""The rows either side of the hot path timed on the MI355X, each beside a CPU baseline
on the same box's host cores:

  graph construction  featurize.protein_graph (eqd_protein_graph_*) for 200 / 1 000 / 2 000-residue proteins.  CPU baseline:
                      the reference's own compute_dig_kNN_graph loop cannot travel (src/utils/protein_utils.py:311-397; its
                      timing in the build container is profiles/r03_graph_reference_cpu.txt) - here the same O(N^2) loop of
                      scipy cdist calls over all-atom coordinates restated inline (the dominant cost of the reference
                      function), on a bounded sample of rows, extrapolated to N^2 / 2 pairs.
  clash removal       inference.remove_clashes (eqd_clash_iterations), iterations/s at 2 000 x 3 000 atoms.  CPU baseline:
                      the reference's loop body (src/inference_rigid.py:213-232) as restated by oracle/loss_port.py's
                      body_intersection_loss under torch autograd on the host, a bounded number of iterations.
  pocket OT           losses.pocket_ot_loss (device cost / value / gradient + exact host solver, ONE round trip per batch) at
                      configs B and C, ms per batch fwd + bwd.  CPU baseline: the per-pair loop of the reference
                      (src/train.py:117-129) with the oracle's LP solver in POT's place (POT is absent: "POT parity
                      unpinned") - baseline only, an LP solve is slower than POT's network simplex.

usage (GPU box): python profiles/bench_frows.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from equidock_public_amd import featurize as FZ, inference as INF, losses, synthetic  # noqa: E402

SMALL = os.environ.get('EQD_FROWS_SMALL') == '1'      # dry run of this script on the x86 simulator (no GPU): tiny sizes
if SMALL:
    from equidock_public_amd import _lib
    from tests.hostsim import build as _hs
    _lib.load_library_for_testing(_hs.build())
    torch.cuda.synchronize = lambda: None
dev = torch.device('cpu' if SMALL else 'cuda:0')


def synthetic_residues(n, seed):
    """n residues with N / CA / C backbone atoms and 3-9 further atoms each (sizes like a real protein: ~8 atoms per
    residue), C-alpha trace from synthetic._chain"""
    rng = np.random.default_rng(seed)
    ca = synthetic._chain(n, rng)
    res = []
    for i in range(n):
        def unit():
            v = rng.normal(size=3)
            return v / np.linalg.norm(v)
        extra = int(rng.integers(3, 10))
        coords = [ca[i] + 1.46 * unit(), ca[i], ca[i] + 1.52 * unit()] + [ca[i] + rng.normal(size=3) * 1.5 for _ in range(extra)]
        names = ['N', 'CA', 'C'] + ['CB'] * extra
        res.append(FZ.Residue('A', i + 1, 'ALA', names, ['N', 'C', 'C'] + ['C'] * extra, np.asarray(coords, dtype=np.float32)))
    return res


def timed(fn, reps, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print('== graph construction (one protein; cutoff 30, max_neighbor 10) ==')
import scipy.spatial as spa  # noqa: E402
for n in ((30, 40) if SMALL else (200, 1000, 2000)):
    res = synthetic_residues(n, 7 + n)
    ca = np.stack([r.coords[1] for r in res])
    t_dev = timed(lambda: FZ.protein_graph(res, ca, 30.0, 10, dev), 3)
    # the device part alone (host work - frames, ragged atom array, Kabsch alignment - excluded)
    loc, n_i, u_i, v_i = FZ.local_frames(res, True)
    atoms, off = FZ.atoms_ragged(res)
    t_k = timed(lambda: FZ.knn_graph_device(atoms, off, loc.astype(np.float64), n_i, u_i, v_i, 30.0, 10, dev), 5)
    rows = min(n - 1, max(4, 20000 // n))          # bounded sample of the reference's double loop
    t0 = time.perf_counter()
    for i in range(rows):
        for j in range(i + 1, n):
            np.mean(spa.distance.cdist(res[i].coords, res[j].coords))
    per_pair = (time.perf_counter() - t0) / sum(n - 1 - i for i in range(rows))
    cpu = per_pair * n * (n - 1) / 2
    print(f'{n} residues: protein_graph {t_dev * 1e3:.1f} ms (device kernels + transfers {t_k * 1e3:.2f} ms); reference-style host loop '
          f'~{cpu:.1f} s (extrapolated from {rows} rows, 1 thread): {cpu / t_dev:.0f}x')

rng = np.random.default_rng(3)
n_la, n_ra = (50, 60) if SMALL else (2000, 3000)
print(f'== clash removal: {n_la} ligand x {n_ra} receptor atoms ==')
lig = torch.from_numpy((rng.normal(0, 12, (n_la, 3)) + np.array([14.0, 1.0, -2.0])).astype(np.float32))
rec = torch.from_numpy(rng.normal(0, 14, (n_ra, 3)).astype(np.float32))
ligd, recd = lig.to(dev), rec.to(dev)
iters = 20 if SMALL else 400
out = INF.remove_clashes(ligd, recd, loss_stop=-1.0, max_it=iters, check_every=100)      # never converges: fixed work
torch.cuda.synchronize()
t0 = time.perf_counter()
out = INF.remove_clashes(ligd, recd, loss_stop=-1.0, max_it=iters, check_every=100)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / out['iterations']
from oracle import loss_port as lp  # noqa: E402  (CPU baseline only)
eul, tr = torch.zeros(3, requires_grad=True), torch.zeros(3, requires_grad=True)
n_cpu = 5
t0 = time.perf_counter()
for _ in range(n_cpu):
    th = (INF.get_rot_mat(eul) @ lig.t()).t() + tr
    loss = lp.body_intersection_loss(th, rec, 8.0, 8.0)
    loss.backward()
    eul.grad = tr.grad = None
t_cpu = (time.perf_counter() - t0) / n_cpu
print(f'device {1 / t_dev:.0f} iterations/s ({t_dev * 1e6:.0f} us per iteration, {out["iterations"]} iterations); host torch autograd '
      f'(the reference\'s loop body, {torch.get_num_threads()} threads) {1 / t_cpu:.1f} iterations/s: {t_cpu / t_dev:.0f}x')

print('== pocket OT (fwd + bwd per batch, 50 keypoints) ==')
from oracle import ot_port  # noqa: E402  (CPU baseline only)
for name, B_, npk in ((('tiny', 2, 5),) if SMALL else (('B: 8 pairs', 8, 30), ('C: 64 pairs', 64, 30))):
    gen = torch.Generator().manual_seed(5)
    Yl = (torch.randn(B_, 50, 3, generator=gen) * 10).to(dev).requires_grad_(True)
    Yr = (torch.randn(B_, 50, 3, generator=gen) * 10).to(dev).requires_grad_(True)
    pl = [torch.randn(npk, 3, generator=gen) * 10 for _ in range(B_)]
    pr = [torch.randn(npk, 3, generator=gen) * 10 for _ in range(B_)]

    def ours():
        Yl.grad = Yr.grad = None
        losses.pocket_ot_loss(Yl, Yr, pl, pr).mean().backward()
    t_dev = timed(ours, 10)
    Ylc, Yrc = Yl.detach().cpu(), Yr.detach().cpu()
    t0 = time.perf_counter()
    nb = min(B_, 8)
    for p in range(nb):
        ot_port.pocket_ot_loss(pl[p], pr[p], Ylc[p], Yrc[p])
    t_cpu = (time.perf_counter() - t0) / nb * B_
    print(f'{name}, {npk} pocket points each: device + host solver {t_dev * 1e3:.2f} ms per batch; per-pair host loop with the LP '
          f'oracle ~{t_cpu * 1e3:.0f} ms ({t_cpu / t_dev:.0f}x; POT itself is absent)')
