"""Loss block (MSE + body intersection) on the config-B batch: the library's two launches
against the reference's formulation (per-pair Python loop of torch ops with an (n_l x n_r) matrix per term) on the same
GPU.  usage (GPU box): python profiles/bench_losses.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from equidock_public_amd import graph as G, losses, synthetic
from oracle import loss_port as lp          # the torch formulation, used here as the baseline being timed

dev = torch.device('cuda:0')
for name, sizes in (('B: 8 x (200,200)', [(200, 200)] * 8), ('C: 64 x (300,300)', [(300, 300)] * 64)):
    g = G.batch_pairs(synthetic.make_pairs(sizes, 1)).to(dev)
    gen = torch.Generator().manual_seed(0)
    preds = [torch.randn(nl, 3, generator=gen) * 8 for nl, _ in sizes]
    tgts = [a + torch.randn(a.shape, generator=gen) for a in preds]
    recs = [torch.randn(nr, 3, generator=gen) * 9 + 3 for _, nr in sizes]
    pd, td, rd = torch.cat(preds).to(dev), torch.cat(tgts).to(dev), torch.cat(recs).to(dev)
    lp_d, lt_d, lr_d = [a.to(dev) for a in preds], [a.to(dev) for a in tgts], [a.to(dev) for a in recs]

    def ours():
        a = pd.clone().requires_grad_(True)
        m, i = losses.pair_losses(g, a, td, rd, 25.0, 10.0)
        (m.mean() + i.mean()).backward()
        return a.grad

    def torch_loop():
        leaves = [a.clone().requires_grad_(True) for a in lp_d]
        m, i = lp.pair_losses(leaves, lt_d, lr_d, 25.0, 10.0)
        (m.mean() + i.mean()).backward()
        return torch.cat([a.grad for a in leaves])
    ga, gb = ours(), torch_loop()
    err = float((ga - gb).abs().max() / gb.abs().max())
    res = {}
    for fn in (ours, torch_loop):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[fn.__name__] = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: HIP pair_losses fwd+bwd {res['ours']:.3f} ms, torch per-pair loop {res['torch_loop']:.3f} ms "
          f"({res['torch_loop'] / res['ours']:.1f}x); max rel grad difference {err:.2e}")
