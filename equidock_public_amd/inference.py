"""Inference post-processing and evaluation of the drop-in: what src/inference_rigid.py does
after the model call (:199-239) and what the evaluation harness computes (src/utils/eval.py, src/test_all_methods/
eval_pdb_outputset.py:80-109).

    rotation, translation = rotation_list[0], translation_list[0]                 # model outputs (:199-200)
    new_pos = apply_rigid(rotation, translation, ligand_all_atoms)                 # :205
    out = remove_clashes(new_pos, receptor_all_atoms)                              # :207-234, on the device
    write_pdb_coordinates(ligand_pdb, out['positions'], out_path)                  # :237-239

The clash-removal loop - up to 2000 iterations of autograd over an (n_lig_atoms x n_rec_atoms) matrix on the host in the
reference - runs on the MI355X with all of its state resident (eqd_clash_iterations); the host looks at the stop flag every
`check_every` iterations.  No CPU fallback for it.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .featurize import rigid_transform_kabsch_3d


class EqdClashState(C.Structure):
    _fields_ = [('euler', C.c_float * 3), ('trans', C.c_float * 3), ('loss', C.c_float), ('it', C.c_int32),
                ('done', C.c_int32), ('reserved', C.c_int32)]


def get_rot_mat(euler_angles):
    """src/inference_rigid.py:46-73: R = RZ(yaw) RY(pitch) RX(roll) for euler_angles = (roll, yaw, pitch)."""
    roll, yaw, pitch = euler_angles[0], euler_angles[1], euler_angles[2]
    zero, one = torch.zeros_like(roll), torch.ones_like(roll)
    RX = torch.stack([torch.stack([one, zero, zero]), torch.stack([zero, torch.cos(roll), -torch.sin(roll)]),
                      torch.stack([zero, torch.sin(roll), torch.cos(roll)])]).reshape(3, 3)
    RY = torch.stack([torch.stack([torch.cos(pitch), zero, torch.sin(pitch)]), torch.stack([zero, one, zero]),
                      torch.stack([-torch.sin(pitch), zero, torch.cos(pitch)])]).reshape(3, 3)
    RZ = torch.stack([torch.stack([torch.cos(yaw), -torch.sin(yaw), zero]), torch.stack([torch.sin(yaw), torch.cos(yaw), zero]),
                      torch.stack([zero, zero, one])]).reshape(3, 3)
    return torch.mm(torch.mm(RZ, RY), RX)


def apply_rigid(rotation, translation, coords):
    """(rotation @ coords.T).T + translation (src/inference_rigid.py:202, 205) on the tensors' device."""
    R = torch.as_tensor(rotation, dtype=torch.float32, device=coords.device).reshape(3, 3)
    t = torch.as_tensor(translation, dtype=torch.float32, device=coords.device).reshape(1, 3)
    return (R @ coords.to(torch.float32).t()).t() + t


def remove_clashes(ligand_atoms, receptor_atoms, sigma=8.0, surface_ct=8.0, loss_stop=0.5, max_it=2000, check_every=50):
    """src/inference_rigid.py:207-234 on the device.  ligand_atoms [n, 3]: the docked ligand (all atoms, after
    apply_rigid); receptor_atoms [m, 3].  Returns dict(positions [n, 3] device tensor, euler (3,), translation (3,),
    iterations, loss) - `positions` = get_rot_mat(euler) @ ligand_atoms + translation with the FINAL parameters, i.e. like
    the reference's `ligand_th` one gradient step past the evaluation that met the stop rule (:226-232); `loss` is the last
    evaluated loss (`non_int_loss_item`), `iterations` the reference's `it`."""
    lib = _lib.load_library()
    lig = _lib.require_device(ligand_atoms.detach().to(torch.float32).contiguous(), 'ligand atoms')
    rec = _lib.require_device(receptor_atoms.detach().to(torch.float32).contiguous(), 'receptor atoms')
    dev = lig.device
    n, m = lig.shape[0], rec.shape[0]
    wsb = lib.eqd_clash_workspace_bytes(n, m)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    state = torch.zeros(C.sizeof(EqdClashState), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    host = EqdClashState()
    while True:
        with _lib.device_guard(dev):
            _lib.check(lib.eqd_clash_iterations(int(check_every), n, m, _lib.ptr(lig), _lib.ptr(rec), C.c_float(sigma),
                                                C.c_float(surface_ct), C.c_float(loss_stop), int(max_it), _lib.ptr(state),
                                                _lib.ptr(ws), C.c_size_t(wsb), st))
        raw = state.cpu().numpy().tobytes()            # the periodic look at the stop flag
        C.memmove(C.byref(host), raw, C.sizeof(EqdClashState))
        if host.done:
            break
    euler, trans = np.asarray(host.euler[:], dtype=np.float32), np.asarray(host.trans[:], dtype=np.float32)
    pos = apply_rigid(get_rot_mat(torch.from_numpy(euler)).to(dev), trans, lig)      # :230
    return {'positions': pos, 'euler': euler, 'translation': trans, 'iterations': int(host.it), 'loss': float(host.loss)}


def write_pdb_coordinates(src_pdb, coords, out_pdb):
    """Rewrite the ATOM records of `src_pdb` with new coordinates (row order = ATOM order, as the reference assigns them
    into biopandas' ATOM table and saves records=['ATOM'], src/inference_rigid.py:237-239)."""
    coords = np.asarray(coords.detach().cpu() if torch.is_tensor(coords) else coords, dtype=np.float64)
    k = 0
    with open(src_pdb) as f, open(out_pdb, 'w') as o:
        for line in f:
            if line.startswith('ATOM'):
                x, y, z = coords[k]
                line = f"{line[:30]}{x:8.3f}{y:8.3f}{z:8.3f}{line[54:]}"
                k += 1
                o.write(line if line.endswith('\n') else line + '\n')
    if k != len(coords):
        raise ValueError(f"{src_pdb} has {k} ATOM records for {len(coords)} coordinate rows")


def read_pdb_atoms(path, ca_only=False):
    """ATOM coordinates in file order (get_nodes_coors_numpy, src/inference_rigid.py:150-154)."""
    rows = []
    with open(path) as f:
        for line in f:
            if line.startswith('ATOM') and (not ca_only or line[12:16].strip() == 'CA'):
                rows.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    return np.asarray(rows, dtype=np.float32).reshape(-1, 3)


# ---- evaluation (src/utils/eval.py:12-77, src/test_all_methods/eval_pdb_outputset.py:80-109) ------------------------
def rmsd_metrics(ligand_pred, receptor_pred, ligand_true, receptor_true):
    """Meter_Unbound_Bound.update_rmsd: (ligand RMSD, receptor RMSD, complex RMSD after Kabsch alignment of the predicted
    complex onto the true one)."""
    a = [np.asarray(t.detach().cpu() if torch.is_tensor(t) else t, dtype=np.float32) for t in
         (ligand_pred, receptor_pred, ligand_true, receptor_true)]
    lp, rp, lt, rt = a
    lig = np.sqrt(np.mean(np.sum((lp - lt) ** 2, axis=1)))
    rec = np.sqrt(np.mean(np.sum((rp - rt) ** 2, axis=1)))
    cp, ctrue = np.concatenate((lp, rp), axis=0), np.concatenate((lt, rt), axis=0)
    R, b = rigid_transform_kabsch_3d(cp.T, ctrue.T)
    aligned = ((R @ cp.T) + b).T
    return lig, rec, np.sqrt(np.mean(np.sum((aligned - ctrue) ** 2, axis=1)))


class Meter_Unbound_Bound:
    """src/utils/eval.py:12-77 (same method names)."""

    def __init__(self):
        self.complex_rmsd_list, self.ligand_rmsd_list, self.receptor_rmsd_list = [], [], []

    def update_rmsd(self, ligand_coors_pred, receptor_coors_pred, ligand_coors_true, receptor_coors_true):
        lig, rec, cpx = rmsd_metrics(ligand_coors_pred, receptor_coors_pred, ligand_coors_true, receptor_coors_true)
        self.complex_rmsd_list.append(cpx)
        self.ligand_rmsd_list.append(lig)
        self.receptor_rmsd_list.append(rec)
        return cpx

    def summarize(self, reduction_rmsd='median'):
        if reduction_rmsd not in ('mean', 'median'):
            raise ValueError("Meter_Unbound_Bound: reduction_rmsd mis specified!")
        f = np.mean if reduction_rmsd == 'mean' else np.median
        return f(np.array(self.ligand_rmsd_list)), f(np.array(self.receptor_rmsd_list)), f(np.array(self.complex_rmsd_list))

    def summarize_with_std(self, reduction_rmsd='median'):
        if reduction_rmsd not in ('mean', 'median'):
            raise ValueError("Meter_Unbound_Bound: reduction_rmsd mis specified!")
        arr = np.array(self.complex_rmsd_list)
        return (np.mean(arr) if reduction_rmsd == 'mean' else np.median(arr)), np.std(arr)


def complex_and_interface_rmsd(ligand_model_ca, receptor_model_ca, ligand_gt_ca, receptor_gt_ca, cutoff=8.0):
    """CRMSD and IRMSD of one complex (src/test_all_methods/eval_pdb_outputset.py:80-100): the interface = C-alpha pairs of
    the ground truth closer than `cutoff`."""
    lg, rg = np.asarray(ligand_gt_ca, dtype=np.float64), np.asarray(receptor_gt_ca, dtype=np.float64)
    d = np.sqrt(((lg[:, None, :] - rg[None, :, :]) ** 2).sum(-1))
    al, ar = np.where(d < cutoff)
    crmsd = rmsd_metrics(ligand_model_ca, receptor_model_ca, ligand_gt_ca, receptor_gt_ca)[2]
    irmsd = rmsd_metrics(np.asarray(ligand_model_ca)[al], np.asarray(receptor_model_ca)[ar], np.asarray(ligand_gt_ca)[al],
                         np.asarray(receptor_gt_ca)[ar])[2]
    return crmsd, irmsd
