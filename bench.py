"""bench.py -- protein-pairs/sec (fwd+bwd) of one IEGMN stack on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload B|A|C|E]

A "step" = zero grads -> Rigid_Body_Docking_Net forward -> fixed scalar loss
-> backward, on one synthetic batch that is already resident in HBM (the slice
src/train.py:88-100,154 of the reference without losses/OT/clip/optimizer).  Default workload =
BASELINE.json configs[1]: 8 pairs x (200, 200) residues, k = 10, 8-layer IEGMN, hdim 64, 50
heads, fp32, on ONE GPU.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
every rank runs the same per-GPU batch (weak scaling) and the flat gradient buffer is
all-reduced once per step over RCCL.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     the dominant kernel (edge-message backward/forward), measured live with HIP
                  events on the launch stream: algorithmic FLOPs|bytes per launch / avg duration;
  "cpu_baseline": the oracle (oracle/iegmn_port.py, reference op sequence incl. the dense mask)
                  timed on the host cores on the same batch (N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (pairs per GPU, (n_lig, n_rec), layers, shared, skip_weight_h, description)
    'A': (1, (200, 200), 5, True, 0.5, 'A: DB5.5 single pair (200,200), k=10, 5-layer shared IEGMN hdim 64, 50 heads, fp32'),
    'B': (8, (200, 200), 8, False, 0.75, 'B: DB5.5-sized batch of 8 pairs x (200,200) residues, k=10, 8-layer IEGMN hdim 64, 50 heads, fp32'),
    'C': (64, (300, 300), 8, False, 0.75, 'C-fp32: DIPS-sized batch of 64 pairs x (300,300) residues, k=10, 8-layer IEGMN (fp32 arithmetic)'),
    'E': (4, (2000, 2000), 8, False, 0.75, 'E: stress, 4 pairs x (2000,2000) residues, k=10, 8-layer IEGMN, fp32'),
}
PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md; no sparsity)
PEAK_HBM_GBS = 8000.0      # HBM3E spec
EDGE_FWD_FLOP_PER_EDGE = 38272          # SURVEY.md section 8d, layers >= 1, model as written
EDGE_FWD_BYTES = lambda n, e: n * 540 + e * 112   # noqa: E731  SURVEY.md section 8d (fp32)


class _BatchedLoss(torch.autograd.Function):
    """sum over pairs of mean(lig'^2) + mean(Yl^2) + mean(Yr^2) on the batched outputs (lig_w[i] = 1 / (3 n_pair(i))).

    The same scalar as oracle.iegmn_port.scalar_loss, written as ONE weighted sum of squares over the concatenated
    outputs with the backward spelled out: differentiated by autograd from the obvious expression it is 27 elementwise /
    reduction launches (~100 us of GPU time at workload B, 6 % of the step, none of it the path being measured); this
    way it is 6."""

    @staticmethod
    def forward(ctx, lig, Yl, Yr, lig_w):
        c = 1.0 / (Yl.shape[1] * Yl.shape[2])
        key = (lig_w.data_ptr(), Yl.shape, Yr.shape)
        if _BatchedLoss._wcat is None or _BatchedLoss._wcat[0] != key:      # weights of the concatenation, built once
            w = torch.cat([lig_w.expand(-1, 3).reshape(-1), torch.full((Yl.numel() + Yr.numel(),), c, dtype=lig_w.dtype,
                                                                       device=lig_w.device)])
            _BatchedLoss._wcat = (key, w)
        w = _BatchedLoss._wcat[1]
        v = torch.cat([lig.reshape(-1), Yl.reshape(-1), Yr.reshape(-1)])
        vw = v * w
        ctx.save_for_backward(vw)
        ctx.shapes = (lig.shape, Yl.shape, Yr.shape)
        return (vw * v).sum()

    @staticmethod
    def backward(ctx, g):
        (vw,) = ctx.saved_tensors
        gv = vw * (g * 2.0)
        sl, syl, syr = ctx.shapes
        nl, nyl = sl.numel(), syl.numel()
        return gv[:nl].view(sl), gv[nl:nl + nyl].view(syl), gv[nl + nyl:].view(syr), None


_BatchedLoss._wcat = None


def batched_loss(lig, Yl, Yr, lig_w):
    return _BatchedLoss.apply(lig, Yl, Yr, lig_w)


def time_kernel(fn, iters, stream_sync):
    """Average duration of ONE launch of fn, from HIP events on the launch stream around batches of 10 back-to-back
    launches.  A spin kernel in front of every batch keeps the GPU busy while the host enqueues the batch, so host launch
    overhead (ctypes, Python; 5-15 us per launch depending on the box) is not in the figure - one event pair per launch
    measured 37-41 us for a kernel that rocprofv3 times at 34 us."""
    for _ in range(3):
        fn()
    stream_sync()
    per = 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(1, iters // per))]
    for e0, e1 in ev:
        torch.cuda._sleep(2_000_000)        # ~1 ms of spinning
        e0.record()
        for _ in range(per):
            fn()
        e1.record()
    ev[-1][1].synchronize()
    ts = [e0.elapsed_time(e1) / per for e0, e1 in ev]
    return sum(ts) / len(ts) * 1e-3   # seconds per launch


def edge_kernel_rooflines(net, packed, dev, workload='B', bf16=False):
    """Launch the edge-message kernels standalone (layer-1 weights, the workload's graph) on torch's
    current stream and time them with HIP events on that stream."""
    from equidock_public_amd import _lib
    lib = _lib.load_library()
    gs = _lib.graph_struct(packed)
    lay = net.iegmn_original.iegmn_layers[1]
    N, E = packed.n_nodes, packed.n_edges
    f = dict(dtype=torch.float32, device=dev)
    P, Q = torch.randn(N, 64, **f), torch.randn(N, 64, **f)
    x = packed.x0.clone()
    aggr, xnew = torch.empty(N, 64, **f), torch.empty(N, 3, **f)
    ep = _lib.EqdEdgeParams()
    W1 = lay.edge_mlp[0].weight
    ep.W1, ep.ldw1, ep.d_in = W1.data_ptr(), W1.shape[1], 64
    ep.ln_g, ep.ln_b = lay.edge_mlp[3].weight.data_ptr(), lay.edge_mlp[3].bias.data_ptr()
    ep.W2, ep.b2 = lay.edge_mlp[4].weight.data_ptr(), lay.edge_mlp[4].bias.data_ptr()
    ep.Wc1, ep.bc1 = lay.coors_mlp[0].weight.data_ptr(), lay.coors_mlp[0].bias.data_ptr()
    ep.wc2, ep.bc2 = lay.coors_mlp[4].weight.data_ptr(), lay.coors_mlp[4].bias.data_ptr()
    ep.slope, ep.ln_eps, ep.eta, ep.use_dist, ep.use_he = 0.01, 1e-5, 0.0, 1, 1
    ep.bf16 = int(bf16)
    st = _lib.stream_ptr(dev)
    sync = lambda: torch.cuda.current_stream(dev).synchronize()  # noqa: E731

    def fwd():
        _lib.check(lib.eqd_edge_message_fwd(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q), _lib.ptr(x),
                                            _lib.ptr(aggr), _lib.ptr(xnew), st))
    t_fwd = time_kernel(fwd, 50, sync)

    # backward kernel alone (k_edge_bwd): the C entry point also runs the weight-gradient GEMMs and the
    # CSC gather, so time the entry point and report the kernel's share from the profile separately.
    wsb = lib.eqd_edge_message_bwd_workspace_bytes(C.byref(gs))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dag, dxn = torch.randn(N, 64, **f), torch.randn(N, 3, **f)
    dP, dQ, dx = torch.empty(N, 64, **f), torch.empty(N, 64, **f), torch.empty(N, 3, **f)
    g = {k: torch.zeros_like(v) for k, v in (('W1', W1), ('lng', lay.edge_mlp[3].weight), ('lnb', lay.edge_mlp[3].bias),
                                             ('W2', lay.edge_mlp[4].weight), ('b2', lay.edge_mlp[4].bias),
                                             ('Wc1', lay.coors_mlp[0].weight), ('bc1', lay.coors_mlp[0].bias),
                                             ('wc2', lay.coors_mlp[4].weight), ('bc2', lay.coors_mlp[4].bias))}
    eg = _lib.EqdEdgeGrads()
    eg.dW1, eg.ldw1 = g['W1'].data_ptr(), W1.shape[1]
    eg.dln_g, eg.dln_b, eg.dW2, eg.db2 = (g[k].data_ptr() for k in ('lng', 'lnb', 'W2', 'b2'))
    eg.dWc1, eg.dbc1, eg.dwc2, eg.dbc2 = (g[k].data_ptr() for k in ('Wc1', 'bc1', 'wc2', 'bc2'))

    def bwd_op():
        _lib.check(lib.eqd_edge_message_bwd(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q), _lib.ptr(x),
                                            _lib.ptr(dag), _lib.ptr(dxn), _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dx),
                                            C.byref(eg), _lib.ptr(ws), C.c_size_t(wsb), st))

    def bwd():
        _lib.check(lib.eqd_edge_message_bwd_kernel_only(C.byref(gs), C.byref(ep), _lib.ptr(P), _lib.ptr(Q),
                                                        _lib.ptr(x), _lib.ptr(dag), _lib.ptr(dxn), _lib.ptr(dQ),
                                                        _lib.ptr(dx), _lib.ptr(ws), C.c_size_t(wsb), st))
    t_bwd = time_kernel(bwd, 30, sync)
    t_bwd_op = time_kernel(bwd_op, 20, sync)
    flop_f = EDGE_FWD_FLOP_PER_EDGE * E
    byte_f = EDGE_FWD_BYTES(N, E)
    out = {}
    if bf16:
        byte_f = N * 540 + E * (27 * 2 + 4)       # he rows in bf16
    peak_tf = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
    for name, t, mult in (('k_edge_fwd', t_fwd, 1), ('k_edge_bwd', t_bwd, 2)):
        tf = flop_f * mult / t / 1e12
        gb = byte_f * mult / t / 1e9
        out[name] = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak_tf, "unit": "TFLOP/s",
                     "frac": round(tf / peak_tf, 4), "traffic": None,
                     "avg_launch_us": round(t * 1e6, 2), "algorithmic_flops_per_launch": flop_f * mult,
                     "algorithmic_bytes_per_launch": byte_f * mult,
                     "hbm_algorithmic_GBps": round(gb, 1), "hbm_frac_algorithmic": round(gb / PEAK_HBM_GBS, 4)}
        if bf16 and gb / PEAK_HBM_GBS > tf / peak_tf:
            # with the GEMMs at the bf16 MFMA rate the HBM side is the closer roof of the two
            out[name].update({"bound": "hbm", "achieved": round(gb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(gb / PEAK_HBM_GBS, 4), "mfma_TFLOPs": round(tf, 3),
                              "mfma_frac_bf16_peak": round(tf / peak_tf, 4)})
    out['k_edge_bwd']['whole_bwd_op_us'] = round(t_bwd_op * 1e6, 2)   # + weight-grad GEMMs, reductions, CSC gather
    # HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json; counters cannot be read
    # from inside the process): (2 * FETCH_SIZE + WRITE_SIZE) KB, see the file's comment for the correction
    try:
        tr = {} if bf16 else json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json'))).get(workload, {})
        for k, v in tr.items():
            if k in out:
                out[k]['traffic'] = int((2 * v['FETCH_SIZE_KB'] + v['WRITE_SIZE_KB']) * 1024)
                out[k]['traffic_source'] = 'profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'
    except Exception:
        pass
    return out


def cpu_baseline(args_model, sd, g_cpu, pairs_per_step):
    """Oracle (reference op sequence, dense mask) fwd+bwd on the host cores."""
    from oracle import iegmn_port as port
    raw = port.raw_from_graph(g_cpu)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cores = torch.get_num_threads()

    def step():
        for v in leaves.values():
            v.grad = None
        outs = port.forward(leaves, args_model, raw, faithful=True)
        port.scalar_loss(outs).backward()
    for _ in range(1):
        step()
    times = []
    budget = time.perf_counter() + 25.0
    while len(times) < 5 and (time.perf_counter() < budget or len(times) < 2):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(pairs_per_step / med, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fwd+bwd steps of the same {pairs_per_step}-pair batch (median {med * 1e3:.0f} ms/step), "
                      f"oracle/iegmn_port.py faithful mode, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='B', choices=sorted(WORKLOADS))
    ap.add_argument('--dtype', default='f32', choices=('f32', 'bf16'),
                    help='bf16: edge-message kernels in bf16 mode (he rows + GEMM inputs bf16, fp32 accumulate); the rest of the path stays fp32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every kernel of every step from the host instead of replaying a captured hipGraph of the step (zero-grad, forward, loss, backward; the gradient all-reduce always runs outside the graph).  Same kernels either way; the replay takes the host (torch autograd + ~110 launches, 0.7-1.5 ms depending on the box) off the critical path of a 1.5 ms step')
    ap.add_argument('--graph', action='store_true', help='(default; kept for older command lines)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist
    # EQD_BENCH_ONE_DEVICE=1 + EQD_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow (barriers, max over ranks,
    # rank-0 report, gradient all-reduce) on a box with a single GPU; real runs use one GPU per rank over RCCL
    one_dev = os.environ.get('EQD_BENCH_ONE_DEVICE') == '1'
    dev = torch.device('cuda', 0 if one_dev else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('EQD_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from equidock_public_amd import graph, model, parallel, synthetic
    from oracle import iegmn_port as port   # only for default_args/init_state_dict + the cpu_baseline leg

    ppg, (nl, nr), L, shared, skh, desc = WORKLOADS[a.workload]
    args_model = port.default_args(iegmn_n_lays=L, shared_layers=shared, skip_weight_h=skh, device=dev)
    if a.dtype == 'bf16':
        args_model['hip_storage_dtype'] = 'bf16'
    sd = port.init_state_dict(args_model, seed=0)
    net = model.Rigid_Body_Docking_Net(args_model).to(dev)
    net.load_state_dict(sd)
    pairs = synthetic.make_pairs([(nl, nr)] * ppg, seed=1000 + rank)
    g_cpu = graph.batch_pairs(pairs)
    g = graph.batch_pairs(pairs).to(dev)
    packed = g.pack()
    lig_w = torch.cat([torch.full((n, 1), 1.0 / (3 * n)) for n in packed.lig_counts]).to(dev)
    reducer = parallel.FlatGradAllReduce(net)

    def compute():
        reducer.zero()
        lig, Yl, Yr, T, b = net.forward_batched(g)
        loss = batched_loss(lig, Yl, Yr, lig_w)
        loss.backward()
        return loss

    # Default: capture zero-grad -> forward -> loss -> backward ONCE into a hipGraph and replay it every step (every
    # kernel still runs every step; the RCCL all-reduce of the flat gradient stays outside the graph).  --eager launches
    # from the host instead; a failed capture falls back to that as well.
    graph_mode = 'eager'
    static_loss = None
    cuda_graph = None
    if not a.eager:
        try:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()      # no collective in flight while the stream is capturing
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    compute()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            cuda_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cuda_graph, capture_error_mode='thread_local'):
                static_loss = compute()
            graph_mode = 'hipGraph replay'
        except Exception as e:   # capture not possible: fall back to eager launches (still the HIP path)
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            cuda_graph = None
            torch.cuda.synchronize()

    def step():
        if cuda_graph is not None:
            cuda_graph.replay()
            loss = static_loss
        else:
            loss = compute()
        reducer.reduce()
        return loss

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host_dt = time.perf_counter() - t0      # host-side enqueue time (== dt when the step is launch-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    status = net.iegmn_original.last_svd_status
    svd_bad = int((status != 0).sum().item())

    out = None
    if rank == 0:
        total_pairs = ppg * world * a.steps
        out = {
            "metric": "protein-pairs/sec (fwd+bwd) per IEGMN stack", "value": round(total_pairs / dt, 2),
            "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if a.dtype == 'f32' else "bf16 edge-message GEMMs (fp32 accumulate) + f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": desc, "pairs_per_gpu": ppg, "nodes_per_gpu": packed.n_nodes,
                       "edges_per_gpu": packed.n_edges, "layers": L, "parallelism": f"dp{world}",
                       "weights": "PyTorch default init (seed 0), ROT key/query x40 (SURVEY.md section 8c)",
                       "loss": float(loss.detach()), "svd_guard_pairs": svd_bad,
                       "host_enqueue_ms_per_step": round(host_dt / a.steps * 1e3, 4), "launch_mode": graph_mode},
        }
        if not a.no_roofline:
            rl = edge_kernel_rooflines(net, packed, dev, a.workload, bf16=(a.dtype == 'bf16'))
            dom = max(rl, key=lambda k: rl[k]["avg_launch_us"])
            out["roofline"] = dict(rl[dom], kernel=dom)
            out["roofline_all"] = rl
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(port.default_args(iegmn_n_lays=L, shared_layers=shared,
                                                                 skip_weight_h=skh), sd, g_cpu, ppg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
