"""Bit-for-bit comparison of two builds of the library on the GPU (a restructured kernel that claims "same bits"):
  python profiles/exp_r06_bits.py dump OUT.pt         outputs + flat gradient of seeded training steps (fp32 / bf16, with and
                                                      without dropout; 8 pairs of ~200 residues and a ragged batch) with the
                                                      library EQD_EXP_LIBRARY names (or the shipped one)
  python profiles/exp_r06_bits.py cmp A.pt B.pt       every tensor equal bit for bit?"""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))


def dump(path):
    from tests import parity_common as pc
    from tests.parity_common import G, build_model, cat_out, port, synthetic
    dev = torch.device('cuda:0')
    out = {}
    for name, sizes, layers in (('B', [(200, 200)] * 8, 8), ('ragged', [(33, 47), (152, 229), (7, 340), (300, 61)], 5)):
        for bf16 in (False, True):
            for drop in (0.0, 0.25):
                args = port.default_args(iegmn_n_lays=layers, skip_weight_h=0.75, dropout=drop, device=dev)
                if bf16:
                    args = dict(args, hip_storage_dtype='bf16')
                if drop > 0:
                    args = dict(args, hip_dropout_masks='library')
                net = build_model(args, port.init_state_dict(args, seed=4, rot_scale=10.0), dev)
                net.train(True)
                flat = net.iegmn_original.enable_flat_grads()
                flat.zero_()
                g = G.batch_pairs(synthetic.make_pairs(sizes, 13)).to(dev)
                torch.manual_seed(99)
                outs = net(g, epoch=0)
                port.scalar_loss(outs).backward()
                torch.cuda.synchronize()
                out[(name, bf16, drop)] = ([cat_out(list(o)).detach().cpu() for o in outs], flat.detach().cpu().clone())
    torch.save(out, path)
    print('saved', path, len(out), 'cases')


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        oa, fa = A[k]
        ob, fb = B[k]
        eq = all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(fa, fb)
        bad += not eq
        print(k, 'bit-identical' if eq else f'DIFFERENT (gradient max |diff| {float((fa - fb).abs().max()):.3e})',
              'grad max', float(fa.abs().max()))
    print('ALL BIT-IDENTICAL' if not bad else f'{bad} case(s) differ')
    return bad


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        sys.exit(cmp(sys.argv[2], sys.argv[3]))
