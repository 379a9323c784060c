"""Latency experiment: phase timestamps inside k_edge_fwd / k_edge_bwd on the workload-B graph (or `--workload C`: 64 x
(300, 300); `--bf16`: the bf16 kernels).  Needs the -DEQD_TRACE library (python profiles/exp_trace_linear.py --build).
usage (GPU box): python profiles/exp_trace_edge.py [--workload C] [--bf16]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'profiles', '_exp', 'libeqd_trace.so')
import torch
from equidock_public_amd import _lib as L, graph, synthetic

FWD = {0: 'start', 1: 'weights staged', 2: 'tile begin', 3: 'geometry (src/dst/x)', 4: 'feature tile (he, rbf) in LDS',
       5: 'P/Q gathered', 6: 'stage-1 MFMA (W1cd feat)', 7: 'LeakyReLU + LN stats', 8: 'W2 chain', 9: 'Wc1 chain + coef',
       10: 'message tile in LDS', 11: 'per-node means stored'}
BWD = dict(FWD)
BWD.update({12: 'coordinate path + d_chid', 13: 'slabs 1 stored + sync', 14: 'dWc1 slab GEMM', 15: 'dm chain (Wc1^T) + db2',
            16: 'slabs 2 stored + syncs', 17: 'dW2 slab GEMM', 18: 'da1 chain (W2^T)', 19: 'LN/LeakyReLU backward',
            20: 'dz1 store + slabs 3 + dW1cd GEMM', 21: 'd rbf -> dx_rel + sync', 22: 'partials written'})
del BWD[10], BWD[11]

if __name__ == '__main__':
    lib = L.load_library_for_testing(OUT)
    dev = torch.device('cuda:0')
    sizes = [(300, 300)] * 64 if 'C' in sys.argv else [(200, 200)] * 8
    g = graph.batch_pairs(synthetic.make_pairs(sizes, 1000)).to(dev)
    packed = g.pack()
    gs = L.graph_struct(packed)
    N, E = packed.n_nodes, packed.n_edges
    f = dict(dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    W1, W2, Wc1 = torch.randn(64, 170, **f) * 0.1, torch.randn(64, 64, **f) * 0.1, torch.randn(64, 64, **f) * 0.1
    vecs = [torch.randn(64, **f) * 0.1 for _ in range(5)]
    bc2 = torch.zeros(1, **f)
    ep = L.EqdEdgeParams()
    ep.W1, ep.ldw1, ep.d_in = W1.data_ptr(), 170, 64
    ep.ln_g, ep.ln_b, ep.W2, ep.b2 = vecs[0].data_ptr(), vecs[1].data_ptr(), W2.data_ptr(), vecs[2].data_ptr()
    ep.Wc1, ep.bc1, ep.wc2, ep.bc2 = Wc1.data_ptr(), vecs[3].data_ptr(), vecs[4].data_ptr(), bc2.data_ptr()
    ep.slope, ep.ln_eps, ep.eta, ep.use_dist, ep.use_he = 0.01, 1e-5, 0.0, 1, 1
    ep.bf16 = int('--bf16' in sys.argv)
    P, Q = torch.randn(N, 64, **f), torch.randn(N, 64, **f)
    x = packed.x0.clone()
    aggr, xnew = torch.empty(N, 64, **f), torch.empty(N, 3, **f)
    dag, dxn = torch.randn(N, 64, **f), torch.randn(N, 3, **f)
    wsb = lib.eqd_edge_message_bwd_workspace_bytes(C.byref(gs))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)

    def fwd():
        L.check(lib.eqd_edge_message_fwd(C.byref(gs), C.byref(ep), L.ptr(P), L.ptr(Q), L.ptr(x), L.ptr(aggr), L.ptr(xnew), st))

    def bwd():
        L.check(lib.eqd_edge_message_bwd_kernel_only(C.byref(gs), C.byref(ep), L.ptr(P), L.ptr(Q), L.ptr(x), L.ptr(dag),
                                                     L.ptr(dxn), L.ptr(P), L.ptr(xnew), L.ptr(ws), C.c_size_t(wsb), st))
    for name, fn, names in (('k_edge_fwd', fwd, FWD), ('k_edge_bwd', bwd, BWD)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        buf = (C.c_longlong * 1024)()
        lib.eqd_trace_fetch(buf)
        slots = sorted(names)
        ck = {s: buf[2 * s] for s in slots}
        wl = {s: buf[2 * s + 1] for s in slots}
        tot_c, tot_w = ck[slots[-1]] - ck[slots[0]], wl[slots[-1]] - wl[slots[0]]
        print(f"{name}: wave 0 of workgroup 0: {tot_c} clock64 ticks = {tot_w / 100:.2f} us (clock64 {tot_c / max(tot_w, 1) * 100:.0f} MHz)")
        for a, b in zip(slots[:-1], slots[1:]):
            print(f"   {ck[b] - ck[a]:7d}  -> {names[b]}")
        nb = 256
        st_ = [buf[512 + 2 * i] for i in range(nb) if buf[512 + 2 * i + 1] > buf[512 + 2 * i] > 0]
        en_ = [buf[512 + 2 * i + 1] for i in range(nb) if buf[512 + 2 * i + 1] > buf[512 + 2 * i] > 0]
        if st_:
            t0 = min(st_)
            durs = sorted((e - s_) / 100 for s_, e in zip(st_, en_))
            print(f"   workgroups traced {len(st_)}: last start {(max(st_) - t0) / 100:.2f} us, last end {(max(en_) - t0) / 100:.2f} us; "
                  f"duration min/median/max {durs[0]:.2f}/{durs[len(durs) // 2]:.2f}/{durs[-1]:.2f} us")
        # the trace buffer is not cleared between kernels: zero it through a fresh fetch baseline is not possible, so
        # workgroup slots of the second kernel simply overwrite the first
