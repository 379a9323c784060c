"""Seeded synthetic residue graphs with the reference's feature layout.

The reference builds its graphs from PDB files with biopandas + DGL
(src/utils/protein_utils.py:201-416), neither of which exists on the GPU box, and there
is no network for the datasets.  This generator produces graphs of the same *shape and
feature layout* from a seed:

  * C-alpha trace: self-avoiding random walk, 3.8 A steps, confined to a ball whose volume
    is 135 A^3 per residue (extents of ~40 A at 200 residues, like DB5.5's 1AVX);
  * per-residue orthonormal frame (n, u, v): random;
  * k-NN graph, k = 10, cutoff 30 A, edges listed destination-major exactly like
    compute_dig_kNN_graph (src/utils/protein_utils.py:339-346), int32 endpoints;
  * edge features `he` (E, 27): 15 RBFs exp(-d^2 / 1.5^k) (protein_utils.py:71-86) followed by
    the 12 orientation features p, q, k, t in the destination's frame (protein_utils.py:375-390);
  * `mu_r_norm` (n, 5) with sigma in {1, 2, 5, 10, 30} (protein_utils.py:351-359), clamped to
    [1e-3, 1] because the model takes its log (rigid_docking_model.py:469);
  * residue ids uniform in 0..20, stored as float (n, 1) like the reference's `res_feat`;
  * ligand `new_x`: centred `x` under a random rotation + translation of norm <= 5 A
    (src/utils/db5_data.py:195-204, src/utils/args.py:55).

Everything is numpy on the host; the result is a list of (ligand, receptor) dicts that
`equidock_public_amd.graph.batch_pairs` turns into the batched container.
"""
import numpy as np

K_NEIGHBORS = 10
CUTOFF = 30.0
SIGMAS_MU = np.array([1., 2., 5., 10., 30.], dtype=np.float64)
RBF_SCALES = np.array([1.5 ** k for k in range(15)], dtype=np.float64)


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def _chain(n, rng):
    radius = (n * 135.0 * 3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    radius = max(radius, 6.0)
    pts = np.zeros((n, 3))
    for i in range(1, n):
        best = None
        for _ in range(30):
            d = rng.normal(size=3)
            d *= 3.8 / np.linalg.norm(d)
            cand = pts[i - 1] + d
            if np.linalg.norm(cand) > radius:
                continue
            if i > 1:
                dmin = np.min(np.linalg.norm(pts[:i - 1] - cand, axis=1))
                if dmin < 3.0:
                    if best is None:
                        best = cand
                    continue
            best = cand
            break
        if best is None:  # walk back towards the centre
            d = -pts[i - 1]
            nd = np.linalg.norm(d)
            d = d / nd * 3.8 if nd > 1e-6 else np.array([3.8, 0.0, 0.0])
            best = pts[i - 1] + d
        pts[i] = best
    return pts


def make_protein(n, rng, k=K_NEIGHBORS, cutoff=CUTOFF):
    """One protein graph as a dict of numpy arrays (float32 / int32)."""
    x = _chain(n, rng)
    frames = np.stack([_random_rotation(rng) for _ in range(n)], axis=0)  # rows: n_i, u_i, v_i
    diff = x[:, None, :] - x[None, :, :]
    dist = np.sqrt((diff ** 2).sum(-1))
    np.fill_diagonal(dist, np.inf)

    src_l, dst_l, d_l, mu_l = [], [], [], []
    for i in range(n):
        valid = np.where(dist[i] < cutoff)[0]
        if len(valid) > k:
            valid = np.argsort(dist[i], kind='stable')[:k]
        src_l.append(valid.astype(np.int32))
        dst_l.append(np.full(len(valid), i, dtype=np.int32))
        dv = dist[i, valid]
        d_l.append(dv)
        if len(valid) == 0:
            mu_l.append(np.full(5, 1e-3))
            continue
        logits = -(dv.reshape(1, -1) ** 2) / SIGMAS_MU.reshape(-1, 1)
        logits -= logits.max(axis=1, keepdims=True)
        w = np.exp(logits)
        w /= w.sum(axis=1, keepdims=True)
        dvec = x[i][None, :] - x[valid]
        mean_vec = w @ dvec
        denom = w @ np.linalg.norm(dvec, axis=1)
        mu_l.append(np.linalg.norm(mean_vec, axis=1) / denom)
    src = np.concatenate(src_l) if src_l else np.zeros(0, np.int32)
    dst = np.concatenate(dst_l) if dst_l else np.zeros(0, np.int32)
    dd = np.concatenate(d_l) if d_l else np.zeros(0)
    rbf = np.exp(-(dd[:, None] ** 2) / RBF_SCALES[None, :])
    basis = frames[dst]                                   # (E, 3, 3)
    p = np.einsum('eij,ej->ei', basis, x[src] - x[dst])
    q = np.einsum('eij,ej->ei', basis, frames[src][:, 0, :])
    kk = np.einsum('eij,ej->ei', basis, frames[src][:, 1, :])
    t = np.einsum('eij,ej->ei', basis, frames[src][:, 2, :])
    he = np.concatenate([rbf, p, q, kk, t], axis=1)
    mu = np.clip(np.stack(mu_l, axis=0), 1e-3, 1.0)
    res = rng.integers(0, 21, size=(n, 1)).astype(np.float32)
    return {
        'x': x.astype(np.float32),
        'res_feat': res,
        'mu_r_norm': mu.astype(np.float32),
        'src': src.astype(np.int32),
        'dst': dst.astype(np.int32),
        'he': he.astype(np.float32),
    }


def make_pair(n_lig, n_rec, rng, k=K_NEIGHBORS, cutoff=CUTOFF, translation_interval=5.0):
    lig = make_protein(n_lig, rng, k, cutoff)
    rec = make_protein(n_rec, rng, k, cutoff)
    rot = _random_rotation(rng)
    t = rng.normal(size=3)
    t *= rng.uniform(0.0, translation_interval) / np.linalg.norm(t)
    xl = lig['x'].astype(np.float64)
    lig['new_x'] = ((rot @ (xl - xl.mean(0, keepdims=True)).T).T + t).astype(np.float32)
    return lig, rec


def make_pairs(sizes, seed, k=K_NEIGHBORS, cutoff=CUTOFF):
    """sizes: list of (n_lig, n_rec). Returns list of (ligand_dict, receptor_dict)."""
    rng = np.random.default_rng(seed)
    return [make_pair(nl, nr, rng, k, cutoff) for nl, nr in sizes]


# DB5.5 bound-structure size statistics measured by the survey: used by
# the "realistic sizes" bench variant.
def realistic_sizes(n_pairs, seed):
    rng = np.random.default_rng(seed)
    lig = np.clip(rng.lognormal(np.log(153.0), 0.6, n_pairs), 29, 1500).astype(int)
    rec = np.clip(rng.lognormal(np.log(311.0), 0.5, n_pairs), 40, 2130).astype(int)
    return list(zip(lig.tolist(), rec.tolist()))
