"""Run-to-run determinism of one training step on the GPU: the same seeded step N times in one process, every output and every
parameter gradient compared bit for bit with the first run.
  python profiles/exp_r06_determinism.py [bf16|f32] [dropout] [layers] [pairs] [size] [runs]"""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
from tests.parity_common import G, build_model, cat_out, port, synthetic      # noqa: E402

bf16 = (sys.argv[1] if len(sys.argv) > 1 else 'bf16') == 'bf16'
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
size = int(sys.argv[5]) if len(sys.argv) > 5 else 200
runs = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device('cuda:0')
args = port.default_args(iegmn_n_lays=layers, skip_weight_h=0.75, dropout=drop, device=dev)
if bf16:
    args = dict(args, hip_storage_dtype='bf16')
if drop > 0:
    args = dict(args, hip_dropout_masks=os.environ.get('DET_MASKS', 'library'))
if os.environ.get('DET_POISON', '0') == '1':      # saved-state / scratch workspaces pre-filled with NaN bit patterns before every call
    from equidock_public_amd import model as _M
    _M.POISON_WORKSPACES = True
net = build_model(args, port.init_state_dict(args, seed=4, rot_scale=10.0), dev)
net.train(os.environ.get('DET_EVAL', '0') != '1')
flat = net.iegmn_original.enable_flat_grads()
sizes = [(size, size)] * pairs
if os.environ.get('DET_RAGGED', '0') == '1':      # a ragged batch: `pairs` pairs between 29 and `size` residues
    import random
    rnd = random.Random(7)
    sizes = [(rnd.randint(29, size), rnd.randint(40, size)) for _ in range(pairs)]
g = G.batch_pairs(synthetic.make_pairs(sizes, 13)).to(dev)
names = [n for n, _ in net.named_parameters()]
ref = None
print(f'bf16={bf16} dropout={drop} layers={layers} pairs={pairs} x ({size},{size}) train={net.training}')
for r in range(runs):
    flat.zero_()
    torch.manual_seed(99)
    outs = net(g, epoch=0)
    port.scalar_loss(outs).backward()
    torch.cuda.synchronize()
    cur = ([cat_out(list(o)).detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    if ref is None:
        ref = cur
        continue
    od = [float((a - b).abs().max()) for a, b in zip(cur[0], ref[0])]
    nan = sum(int(not torch.isfinite(v).all()) for v in cur[1].values())
    gd = {n: float((cur[1][n] - ref[1][n]).abs().max()) for n in names if not torch.equal(cur[1][n], ref[1][n])}
    short = {n.replace('iegmn_original.', '').replace('iegmn_layers.', 'L'): f'{v:.1e}' for n, v in gd.items()}
    print(f'run {r}: outputs max |diff| {od}; {len(gd)} of {len(names)} gradient tensors differ, {nan} with NaN', short if len(short) <= 30 else list(short)[:30])
