"""Latency experiment: phase timestamps inside linear_tile (clock64 + wall_clock64) for the backward 'dh' job
shape (6 sources of K = 64 into 64 outputs, 3200 rows) and a 1-source job.  Builds a separate library with
-DEQD_TRACE into profiles/_exp/ (never the product library).  usage (GPU box): python profiles/exp_trace_linear.py"""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'profiles', '_exp', 'libeqd_trace.so')


def build(fine='--fine' in sys.argv):
    """--fine additionally stamps every pipeline step inside linear_tile (this script's own experiment)"""
    srcs = sorted(glob.glob(os.path.join(ROOT, 'equidock_public_amd', 'csrc', '*.hip')))
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-DEQD_TRACE', '-fgpu-rdc',
           '-shared', '-o', OUT] + (['-DEQD_TRACE_FINE'] if fine else []) + srcs
    subprocess.run(cmd, check=True)


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
        sys.exit(0)
    import torch
    from equidock_public_amd import _lib as L
    lib = L.load_library_for_testing(OUT)
    dev = torch.device('cuda:0')
    rows = 3200
    st = L.stream_ptr(dev)
    for nsrc, trans in ((1, 0), (6, 0), (6, 1)):
        Xs = [torch.randn(rows, 64, device=dev) for _ in range(nsrc)]
        Ws = [torch.randn(64, 261, device=dev) for _ in range(nsrc)]
        Y = torch.zeros(rows, 64, device=dev)
        J = L.EqdLinJob()
        J.nsrc = nsrc
        for i in range(nsrc):
            J.s[i].X, J.s[i].W, J.s[i].ldx, J.s[i].K = Xs[i].data_ptr(), Ws[i].data_ptr(), 64, 64
            J.s[i].w_rs, J.s[i].w_cs = (1, 261) if trans else (261, 1)
        J.M, J.rows, J.alpha, J.slope, J.ln_eps, J.Y, J.ldy = 64, rows, 1.0, 0.01, 1e-5, Y.data_ptr(), 64
        for _ in range(5):
            L.check(lib.eqd_linear(C.byref(J), 1, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.check(lib.eqd_linear(C.byref(J), 1, st)); e1.record(); e1.synchronize()
        buf = (C.c_longlong * 1024)()
        lib.eqd_trace_fetch(buf)
        # trace points: start, after the two prologue loads, then per step (top, stored, synced, next loads
        # issued, multiplied), end of the tile
        n = 2 + 5 * nsrc + 1
        ck = [buf[2 * i] for i in range(n)]
        wl = [buf[2 * i + 1] for i in range(n)]
        tot_c, tot_w = ck[-1] - ck[0], wl[-1] - wl[0]
        print(f"nsrc={nsrc} trans={trans}: kernel {e0.elapsed_time(e1)*1e3:.1f} us (event), traced WG: {tot_c} clock64 ticks, "
              f"{tot_w} wall ticks (100 MHz => {tot_w/100:.2f} us) => clock64 rate {tot_c/max(tot_w,1)*100:.0f} MHz")
        d = [ck[i + 1] - ck[i] for i in range(n - 1)]
        print("  prologue loads:", d[0])
        for k in range(nsrc):
            print(f"  step {k}: wait+top {d[1 + 5 * k]}, store {d[2 + 5 * k]}, sync {d[3 + 5 * k]}, load issue {d[4 + 5 * k]}, mma {d[5 + 5 * k]}")
        print("  epilogue:", d[-1])
        # per-workgroup start / end (100 MHz wall clock), relative to the earliest start
        nb = min(256, (rows + 15) // 16)
        st_ = [buf[512 + 2 * i] for i in range(nb)]
        en_ = [buf[512 + 2 * i + 1] for i in range(nb)]
        t0 = min(st_)
        durs = sorted((e - s_) / 100 for s_, e in zip(st_, en_))
        print(f"  workgroups: first start 0, last start {(max(st_) - t0) / 100:.2f} us, last end {(max(en_) - t0) / 100:.2f} us; "
              f"duration min/median/max {durs[0]:.2f}/{durs[len(durs) // 2]:.2f}/{durs[-1]:.2f} us")
