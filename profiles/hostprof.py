import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from equidock_public_amd import graph, model, parallel, synthetic
from oracle import iegmn_port as port
import bench
dev = torch.device('cuda:0')
args = port.default_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev)
net = model.Rigid_Body_Docking_Net(args).to(dev); net.load_state_dict(port.init_state_dict(args, 0))
pairs = synthetic.make_pairs([(200, 200)] * 8, 1000)
g = graph.batch_pairs(pairs).to(dev); packed = g.pack()
lig_w = torch.cat([torch.full((n, 1), 1.0 / (3 * n)) for n in packed.lig_counts]).to(dev)
red = parallel.FlatGradAllReduce(net)
def step():
    red.zero(); lig, Yl, Yr, T, b = net.forward_batched(g); loss = bench.batched_loss(lig, Yl, Yr, lig_w); loss.backward(); red.reduce()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
