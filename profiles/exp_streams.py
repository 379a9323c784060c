"""Experiment (round 2): workload B (8 pairs x (200, 200), 8 layers) as S independent groups of 8 / S pairs on S HIP streams,
captured as ONE hipGraph with S parallel branches.  Every kernel of the step is latency-bound at this size (one tile per
wave, a few hundred workgroups), so independent chains should hide each other's latency.  S model replicas (own flat
gradient buffers) stand in for what would be per-stream gradient buffers of one model.
usage (GPU box): python profiles/exp_streams.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from equidock_public_amd import config, graph as G, losses, model, parallel, synthetic

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
args = config.published_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev)
sd = config.seeded_state_dict(args, 0)
pairs = synthetic.make_pairs([(200, 200)] * 8, 1000)
for S in (1, 2, 4, 8):
    nets, reds, gs, sls, streams = [], [], [], [], []
    per = 8 // S
    for i in range(S):
        n = model.Rigid_Body_Docking_Net(args).to(dev)
        n.load_state_dict(sd)
        nets.append(n)
        reds.append(parallel.FlatGradAllReduce(n))
        g = G.batch_pairs(pairs[i * per:(i + 1) * per]).to(dev)
        gs.append(g)
        sls.append(losses.ScalarLoss(g.pack(), 50))
        streams.append(torch.cuda.Stream())

    def compute():
        cur = torch.cuda.current_stream()
        for i in range(S):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                reds[i].zero()
                lig, Yl, Yr, T, b = nets[i].forward_batched(gs[i])
                loss, grads = sls[i](lig, Yl, Yr)
                torch.autograd.backward([lig, Yl, Yr], list(grads))
        for i in range(S):
            cur.wait_stream(streams[i])
        for i in range(1, S):
            reds[0].flat.add_(reds[i].flat)

    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            compute()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode='thread_local'):
        compute()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        gr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    # eager
    for _ in range(5):
        compute()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        compute()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / 30
    print(f'S = {S}: hipGraph replay {dt * 1e3:.3f} ms/step = {8 / dt:.0f} pairs/s; eager {de * 1e3:.3f} ms/step = {8 / de:.0f} pairs/s; '
          f'grad checksum {float(reds[0].flat.double().abs().sum()):.6e}', flush=True)
