"""ORACLE-ONLY TOOL: tests/golden/inference_case.npz from the REAL reference functions, in THIS container.

src/inference_rigid.py and src/utils/eval.py cannot be imported (argparse at import, biopandas, DGL), so - like the other
generators - `get_rot_mat`, `G_fn`, `compute_body_intersection_loss` (src/inference_rigid.py:33-73),
`rigid_transform_Kabsch_3D` (src/utils/protein_utils.py:31-64) and the class `Meter_Unbound_Bound`
(src/utils/eval.py:12-77) are taken out of the reference's source files with `ast` and executed as they are.  The
clash-removal loop itself is inline code of the reference's main() (:207-234); it is restated below, line for line, around
those extracted functions.  Recorded: a synthetic clashing ligand / receptor, the loop's final coordinates, iteration count
and loss; and CRMSD / IRMSD-style numbers from the reference's meter.

    python oracle/make_golden_inference.py
"""
import ast
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def extract(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns


def clash_loop(ns, unbound_ligand_new_pos, gt_receptor_nodes_coors, max_it=2000):
    """src/inference_rigid.py:207-234 (the reference's own get_rot_mat / compute_body_intersection_loss)."""
    get_rot_mat, inter = ns['get_rot_mat'], ns['compute_body_intersection_loss']
    euler_angles_finetune = torch.zeros([3], requires_grad=True)
    translation_finetune = torch.zeros([3], requires_grad=True)
    ligand_th = (get_rot_mat(euler_angles_finetune) @ torch.from_numpy(unbound_ligand_new_pos).T).T + translation_finetune
    non_int_loss_item = 100.
    it = 0
    losses = []
    while non_int_loss_item > 0.5 and it < max_it:
        non_int_loss = inter(ligand_th, gt_receptor_nodes_coors, sigma=8, surface_ct=8)
        non_int_loss_item = non_int_loss.item()
        losses.append(non_int_loss_item)
        eta = 1e-3
        if non_int_loss < 2.:
            eta = 1e-4
        if it > 1500:
            eta = 1e-2
        non_int_loss.backward()
        translation_finetune = translation_finetune - eta * translation_finetune.grad.detach()
        translation_finetune = translation_finetune.detach().clone().requires_grad_(True)
        euler_angles_finetune = euler_angles_finetune - eta * euler_angles_finetune.grad.detach()
        euler_angles_finetune = euler_angles_finetune.detach().clone().requires_grad_(True)
        ligand_th = (get_rot_mat(euler_angles_finetune) @ torch.from_numpy(unbound_ligand_new_pos).T).T + translation_finetune
        it += 1
    return ligand_th.detach().numpy(), it, losses, euler_angles_finetune.detach().numpy(), translation_finetune.detach().numpy()


def main():
    ns = {'torch': torch, 'np': np, 'math': math}
    extract('/root/reference/src/inference_rigid.py', ['get_rot_mat', 'G_fn', 'compute_body_intersection_loss'], ns)
    extract('/root/reference/src/utils/protein_utils.py', ['rigid_transform_Kabsch_3D'], ns)
    extract('/root/reference/src/utils/eval.py', ['Meter_Unbound_Bound'], ns)
    rng = np.random.default_rng(7)
    out = {}
    # two globular blobs that overlap: the loop has to push the ligand out
    # 'a': a deep clash, first 300 iterations (tight comparison of the trajectory: float32 gradient descent on a
    # non-convex loss drifts apart over thousands of steps whatever the summation order); 'b': a shallow clash that
    # converges THROUGH the stop threshold (and visits the two smaller step sizes); 'c': the deep clash for all 2000
    # iterations (all three step sizes; compared by its loss)
    cases = {'a': (140, 190, 9.0, 300, 5.0, 6.0, 7), 'b': (60, 75, 15.0, 2000, 4.0, 4.0, 11), 'c': (140, 190, 9.0, 2000, 5.0, 6.0, 7)}
    for tag, (nl, nr, shift, cap, sl, sr, seed) in cases.items():
        rng = np.random.default_rng(seed)
        rec = (rng.normal(0, sr, (nr, 3))).astype(np.float32)
        lig = (rng.normal(0, sl, (nl, 3)) + np.array([shift, 1.0, -2.0])).astype(np.float32)
        pos, it, losses, eul, tr = clash_loop(ns, lig, torch.from_numpy(rec), max_it=cap)
        print(tag, 'iterations', it, 'loss', losses[0], '->', losses[-1])
        out.update({f'{tag}_lig': lig, f'{tag}_rec': rec, f'{tag}_pos': pos, f'{tag}_it': it, f'{tag}_max_it': cap,
                    f'{tag}_losses': np.asarray(losses, dtype=np.float32), f'{tag}_euler': eul, f'{tag}_trans': tr})
    rng = np.random.default_rng(7)
    # get_rot_mat at a non-trivial angle triple
    e = torch.tensor([0.3, -1.1, 0.7])
    out['rot_euler'], out['rot_mat'] = e.numpy(), ns['get_rot_mat'](e).numpy()
    # the reference's meter on a perturbed complex
    lt, rt = rng.normal(0, 10, (57, 3)).astype(np.float32), rng.normal(0, 12, (83, 3)).astype(np.float32) + 15
    lp = (lt @ np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], dtype=np.float32).T + 3.0).astype(np.float32)
    rp = rt + rng.normal(0, 0.5, rt.shape).astype(np.float32)
    meter = ns['Meter_Unbound_Bound']()
    c = meter.update_rmsd(torch.tensor(lp), torch.tensor(rp), torch.tensor(lt), torch.tensor(rt))
    out.update({'m_lp': lp, 'm_rp': rp, 'm_lt': lt, 'm_rt': rt, 'm_complex': c, 'm_ligand': meter.ligand_rmsd_list[0],
                'm_receptor': meter.receptor_rmsd_list[0]})
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'inference_case.npz'), **out)
    print('written')


if __name__ == '__main__':
    main()
