"""Latency experiment: phase timestamps inside k_rowres (default; EQD_ROWWAVE=2) / sub-step timestamps inside k_rowwave (one wave per 16-row tile), taken from a full model step with
the -DEQD_TRACE library (python profiles/exp_trace_linear.py --build).  The LAST k_rowwave launch of a step is the backward
chain of layer 1 (dh of layer 2: 6 sources, d a1n, LayerNorm backward, 3 input gradients = 20 sub-steps of 32 columns).
usage (GPU box): python profiles/exp_trace_rowwave.py [B|C]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'profiles', '_exp', 'libeqd_trace.so')
import torch
from equidock_public_amd import _lib as L, graph, model, synthetic
from oracle import iegmn_port as port

if __name__ == '__main__':
    lib = L.load_library_for_testing(OUT)
    dev = torch.device('cuda:0')
    big = 'C' in sys.argv[1:]
    args = port.default_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev, use_mean_node_features=True,
                             **({'hip_storage_dtype': 'bf16'} if 'bf16' in sys.argv[1:] else {}))
    net = model.Rigid_Body_Docking_Net(args).to(dev)
    net.load_state_dict(port.init_state_dict(args, 0))
    g = graph.batch_pairs(synthetic.make_pairs([(300, 300)] * 64 if big else [(200, 200)] * 8, 1000)).to(dev)
    buf = (C.c_longlong * 1024)()

    def show_res(title, first, n):
        """k_rowres: per source 5 stamps: source begin | rows taken (copies waited) | loads of the next source issued |
        MFMAs done | next weights in LDS (their loads waited) -> barrier -> next source"""
        lib.eqd_trace_fetch(buf)
        print(title)
        print('  source | wait rows + copy | issue next loads | MFMAs | wait + store next weights | barrier + descriptor')
        t0 = buf[2 * (100 + 5 * first)]
        for i in range(first, n):
            a = [buf[2 * (100 + 5 * i + k)] for k in range(6)]
            if a[0] == 0 or a[1] < a[0]:
                break
            nxt = a[5] - a[4] if a[5] > a[4] else -1
            print(f'   {i:3d}: {a[1] - a[0]:6d} | {a[2] - a[1]:6d} | {a[3] - a[2]:6d} | {a[4] - a[3]:6d} | {nxt:6d}     (+{a[0] - t0})')

    def show(title, first):
        lib.eqd_trace_fetch(buf)
        print(title)
        print('  sub-step | descriptor + next loads issued | operands ready + MFMAs | to next sub-step (epilogue / LN backward after a job)')
        t0 = buf[2 * (100 + 3 * first)]
        for i in range(first, 24):
            a, b_, c, d = (buf[2 * (100 + 3 * i + k)] for k in range(4))
            if a == 0 or b_ < a:
                break
            print(f'   {i:3d}: {b_ - a:6d} | {c - b_:6d} | {d - c if d > c else -1:6d}      (+{a - t0})')

    res = os.environ.get('EQD_ROWWAVE', '2') == '2'
    for _ in range(3):
        lig, Yl, Yr, T, b = net.forward_batched(g)
        torch.cuda.synchronize()
        if _ == 2 and res:
            show_res('k_rowres, forward: the head job overwrote source 0; sources 1.. are the last layer node update (h | aggr | att | h0 | a1n):', 1, 6)
        elif _ == 2:
            # forward only: the head's one-source job overwrote sub-steps 0-1; 2.. are the last layer's node update
            # (node_mlp.0: h, aggr_msg, aggr_cross 64 wide + h0 69 wide -> LayerNorm -> node_mlp.4), k-contiguous weights
            show('k_rowwave, forward node-update chain of the last layer (sub-steps 2..):', 2)
        (lig.square().sum() + Yl.square().sum()).backward()
    torch.cuda.synchronize()
    if res:
        show_res('k_rowres, backward chain of layer 1 (dh: 6 sources | Wn2^T | 3 input gradients):', 0, 10)
    else:
        show('k_rowwave, backward chain of layer 1 (m-contiguous weights):', 0)
    wg = [(buf[512 + 2 * i], buf[512 + 2 * i + 1]) for i in range(8)]
    print('wall_clock64 of workgroups 0..7 (start, end):', wg)
