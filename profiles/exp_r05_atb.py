"""A/B of eqd_atb on the weight-gradient jobs of one backward pass at 64 x (300, 300) (38 400 rows, 8 layers x 10 units) with
the Y operands a1n / aggr_msg / h as fp32 tensors (round 4) or as saved bf16 tensors (EqdAtbJob.y_bf16, bf16 storage mode):
same process, same box, HIP events around batches of launches."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from equidock_public_amd import _lib as L
lib = L.load_library()
dev = torch.device('cuda:0')
N, Lyr = 38400, 8
torch.manual_seed(0)
f = dict(device=dev, dtype=torch.float32)
def mk(w): return torch.randn(N, w, **f)
keep = []
def jobs(ybf):
    out = []
    for l in range(Lyr):
        dH, dz, dP, dQ, dq, dk, dv = (mk(64) for _ in range(7))
        a1n, am, ac, h0, h = mk(64), mk(64), mk(64), mk(69), mk(64)
        keep.extend([dH, dz, dP, dQ, dq, dk, dv, a1n, am, ac, h0, h])
        def J(X, Y, n, ybf_ok):
            j = L.EqdAtbJob()
            o = torch.zeros(64, n, **f); keep.append(o)
            if ybf and ybf_ok:
                Yb = Y.to(torch.bfloat16).contiguous(); keep.append(Yb)
                j.Y, j.ldy, j.y_bf16 = Yb.data_ptr(), n, 1
            else:
                j.Y, j.ldy = Y.data_ptr(), n
            j.X, j.ldx, j.M, j.N, j.rows = X.data_ptr(), 64, 64, n, N
            j.out, j.o_rs, j.o_cs, j.slope, j.scale, j.bf16 = o.data_ptr(), n, 1, 0.01, 1.0, 1
            return j
        out += [J(dH, a1n, 64, True), J(dz, am, 64, True), J(dz, ac, 64, False), J(dz, h0, 69, False), J(dz, h, 64, True),
                J(dP, h, 64, True), J(dQ, h, 64, True), J(dq, h, 64, True), J(dk, h, 64, True), J(dv, h, 64, True)]
    return out
res = {}
for ybf in (False, True, False, True):
    js = jobs(ybf)
    arr = (L.EqdAtbJob * len(js))(*js)
    nb = lib.eqd_atb_partial_bytes(arr, len(js))
    part = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = L.stream_ptr(dev)
    for _ in range(3):
        L.check(lib.eqd_atb(arr, len(js), L.ptr(part), C.c_size_t(nb), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(lib.eqd_atb(arr, len(js), L.ptr(part), C.c_size_t(nb), st))
    e1.record(); torch.cuda.synchronize()
    print(f"y_bf16={int(ybf)}: eqd_atb (k_atb + k_atb_reduce) {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call", flush=True)
    keep.clear()
