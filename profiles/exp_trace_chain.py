"""Latency experiment: job boundaries inside the backward row chain of one layer (workload B model), taken from a full
model backward with the -DEQD_TRACE library.  usage (GPU box): python profiles/exp_trace_chain.py"""
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
    args = port.default_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev, use_mean_node_features='--d69' in sys.argv)   # default: every layer 64 wide, so that the traced (last) chain is a typical one
    net = model.Rigid_Body_Docking_Net(args).to(dev)
    net.load_state_dict(port.init_state_dict(args, 0))
    g = graph.batch_pairs(synthetic.make_pairs([(200, 200)] * 8, 1000)).to(dev)
    for _ in range(3):
        lig, Yl, Yr, T, b = net.forward_batched(g)
        (lig.square().sum() + Yl.square().sum()).backward()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 1024)()
    lib.eqd_trace_fetch(buf)
    # the LAST chain launch of the pass wrote the slots: the backward chain of layer 0 (6 jobs: dh of layer 1, da1n,
    # LayerNorm backward, 3 input gradients)
    ck = [buf[2 * s] for s in range(200, 208)]
    names = ['LDS tiles zeroed', 'dh of the layer above (6 sources)', 'da1n = alpha dH Wn2', 'LeakyReLU/LayerNorm backward',
             'd aggr_msg', 'd aggr_cross', 'd h0']
    print('backward row chain, workgroup 0 (clock64 ticks between job boundaries):')
    for i, n in enumerate(names):
        print(f'   {ck[i + 1] - ck[i]:7d}  {n}')
    print(f'   total {ck[6] - ck[0]} ticks')
    print('inside the linear jobs: prologue (epilogue operands requested, step list) | steps | next-job prefetch | epilogue + end sync')
    for jj in (0, 1, 3, 4, 5):
        b = [buf[2 * (210 + 4 * jj + i)] for i in range(4)]
        print(f'   job {jj}: {b[1] - b[0]:6d} | {b[2] - b[1]:6d} | {b[3] - b[2]:6d} | {buf[2 * (201 + jj)] - b[3]:6d}')
