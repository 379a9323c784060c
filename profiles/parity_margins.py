"""Report how much of each tolerance the whole-model golden cases use on the GPU (err / allowed), so that a kernel
change which only re-orders fp32 sums can be judged against the remaining margin.  usage (GPU box):
python profiles/parity_margins.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import parity_common as pc

rows = []
_close, _gclose = pc.close, pc.grad_close


def close(got, ref, tol=1e-4, what=''):
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = max(1.0, float(r.abs().max()))
    rows.append((what, float((g - r).abs().max()), tol * scale))


def grad_close(got, ref, what='', l2=1e-3, mx=1e-2):
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    if float(r.norm()) < 1e-12:
        return
    rows.append((what + ' (rel-L2)', float((g - r).norm()) / float(r.norm()), l2))
    rows.append((what + ' (max/max)', float((g - r).abs().max()) / float(r.abs().max()), mx))


pc.close, pc.grad_close = close, grad_close
dev = torch.device('cuda:0')
for name in ('A_b1_shared5', 'B_b3_dips8', 'C_b2_200', 'D_degraded3', 'E_svd_guard'):
    rows.clear()
    try:
        pc.check_model_case(dev, name)
    except Exception as e:     # noqa: BLE001
        print(name, 'raised', type(e).__name__, e)
    worst = sorted(rows, key=lambda r: -r[1] / r[2])[:4]
    print(name, '  '.join(f"{w}: {e:.2e}/{t:.0e} ({100 * e / t:.0f}%)" for w, e, t in worst))
