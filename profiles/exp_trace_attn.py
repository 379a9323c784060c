"""Latency experiment: phase timestamps inside k_attn_fwd on the workload-B graph (needs the -DEQD_TRACE library:
python profiles/exp_trace_linear.py --build).  usage (GPU box): python profiles/exp_trace_attn.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'profiles', '_exp', 'libeqd_trace.so')
import torch
from equidock_public_amd import _lib as L, graph, synthetic

FWD = {0: 'start', 1: 'K/V loads issued, LDS zeroed, sync', 2: 'Q tile staged, sync', 3: 'Q fragments in registers',
       4: 'K/V tile written to LDS', 5: 'next K/V loads issued', 6: 'S = K Q^T (64 MFMA)', 7: 'online softmax',
       8: 'O += V^T P (64 MFMA)', 9: 'tile loop done', 10: 'merge + store'}

if __name__ == '__main__':
    lib = L.load_library_for_testing(OUT)
    dev = torch.device('cuda:0')
    g = graph.batch_pairs(synthetic.make_pairs([(200, 200)] * 8, 1000)).to(dev)
    packed = g.pack()
    gs = L.graph_struct(packed)
    N = packed.n_nodes
    f = dict(dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    q, k, v = (torch.randn(N, 64, **f) * 0.3 for _ in range(3))
    out, lse = torch.empty(N, 64, **f), torch.empty(N, **f)
    st = L.stream_ptr(dev)
    for _ in range(5):
        L.check(lib.eqd_cross_attention_fwd(C.byref(gs), 64, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), st))
    torch.cuda.synchronize()
    buf = (C.c_longlong * 1024)()
    lib.eqd_trace_fetch(buf)
    slots = sorted(FWD)
    ck = {s: buf[2 * s] for s in slots}
    wl = {s: buf[2 * s + 1] for s in slots}
    print(f"k_attn_fwd: wave 0 of workgroup 0: {ck[10] - ck[0]} clock64 ticks = {(wl[10] - wl[0]) / 100:.2f} us "
          f"(tile phases 4..8 show the LAST of the wave's tiles)")
    for a, b in zip(slots[:-1], slots[1:]):
        print(f"   {ck[b] - ck[a]:7d}  -> {FWD[b]}")
    st_ = [buf[512 + 2 * i] for i in range(256) if buf[512 + 2 * i + 1] > buf[512 + 2 * i] > 0]
    en_ = [buf[512 + 2 * i + 1] for i in range(256) if buf[512 + 2 * i + 1] > buf[512 + 2 * i] > 0]
    t0 = min(st_)
    durs = sorted((e - s_) / 100 for s_, e in zip(st_, en_))
    print(f"   workgroups traced {len(st_)}: last start {(max(st_) - t0) / 100:.2f} us, last end {(max(en_) - t0) / 100:.2f} us; "
          f"duration min/median/max {durs[0]:.2f}/{durs[len(durs) // 2]:.2f}/{durs[-1]:.2f} us")
