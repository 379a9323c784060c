"""Host-side floor of one step: same model (8 layers), a tiny graph so that GPU time is negligible."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from equidock_public_amd import graph, model, parallel, synthetic
from oracle import iegmn_port as port
import bench
dev = torch.device('cuda:0')
args = port.default_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev)
net = model.Rigid_Body_Docking_Net(args).to(dev); net.load_state_dict(port.init_state_dict(args, 0))
for sizes in ([(24, 30)], [(200, 200)] * 8):
    pairs = synthetic.make_pairs(sizes, 1000)
    g = graph.batch_pairs(pairs).to(dev); packed = g.pack()
    lig_w = torch.cat([torch.full((n, 1), 1.0 / (3 * n)) for n in packed.lig_counts]).to(dev)
    red = parallel.FlatGradAllReduce(net)
    def step():
        red.zero(); lig, Yl, Yr, T, b = net.forward_batched(g); loss = bench.batched_loss(lig, Yl, Yr, lig_w); loss.backward(); red.reduce()
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    # forward only
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(100): net.forward_batched(g)
    tf = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"sizes {sizes[0]} x{len(sizes)}: host enqueue {th*10:.3f} ms/step, wall {tt*10:.3f} ms/step, fwd-only host {tf*10:.3f} ms")
