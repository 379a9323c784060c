"""Latency experiment: phases of one partner tile inside the dq pass of k_attn_bwd for a 2000 + 2000 residue pair
(-DEQD_TRACE library: python profiles/exp_trace_linear.py --build).  usage (GPU box): python profiles/exp_trace_attn_bwd.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'profiles', '_exp', 'libeqd_trace.so')
import torch
from equidock_public_amd import _lib as L, graph, synthetic

if __name__ == '__main__':
    lib = L.load_library_for_testing(OUT)
    dev = torch.device('cuda:0')
    g = graph.batch_pairs(synthetic.make_pairs([(2000, 2000)], 1000)).to(dev)
    packed = g.pack()
    gs = L.graph_struct(packed)
    N = packed.n_nodes
    f = dict(dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(N, 64, **f) * 0.3 for _ in range(4))
    out, lse = torch.empty(N, 64, **f), torch.empty(N, **f)
    dq, dk, dv, delta = torch.empty(N, 64, **f), torch.empty(N, 64, **f), torch.empty(N, 64, **f), torch.empty(N, **f)
    st = L.stream_ptr(dev)
    L.check(lib.eqd_cross_attention_fwd(C.byref(gs), 64, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(4):
        if i == 3:
            e0.record()
        L.check(lib.eqd_cross_attention_bwd(C.byref(gs), 64, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(do),
                                            L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(delta), st))
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 1024)()
    lib.eqd_trace_fetch(buf)
    ck = [buf[2 * s] for s in range(30, 35)]
    names = ['next K/V tile loads issued', 'S = K Q^T and dP = V dO^T (128 MFMA)', 'p = exp(S - lse), dS = p (dP - delta)', 'dQ += K^T dS (64 MFMA)']
    print(f'k_attn_bwd, one 2000+2000 pair: {e0.elapsed_time(e1) * 1e3:.1f} us; dq pass, last tile of wave 0 of workgroup 0:')
    for i, n in enumerate(names):
        print(f'   {ck[i + 1] - ck[i]:7d}  {n}')
