"""Where the host time of one step goes: C library calls vs torch/autograd/Python around them (tiny graph: GPU idle)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from equidock_public_amd import graph, model, parallel, synthetic, _lib
from oracle import iegmn_port as port
import bench
dev = torch.device('cuda:0')
args = port.default_args(iegmn_n_lays=8, skip_weight_h=0.75, device=dev)
net = model.Rigid_Body_Docking_Net(args).to(dev); net.load_state_dict(port.init_state_dict(args, 0))
lib = _lib.load_library()
acc = {'fwd_c': 0.0, 'bwd_c': 0.0}
class Timed:
    def __init__(self, fn, key): self.fn, self.key = fn, key
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); acc[self.key] += time.perf_counter() - t0; return r
class LibProxy:
    def __init__(self, lib):
        self._lib = lib
        self.eqd_model_forward = Timed(lib.eqd_model_forward, 'fwd_c')
        self.eqd_model_backward = Timed(lib.eqd_model_backward, 'bwd_c')
    def __getattr__(self, k): return getattr(self._lib, k)
proxy = LibProxy(lib)
_lib.load_library = lambda: proxy
_lib._lib = proxy
for sizes in ([(24, 30)], [(200, 200)] * 8):
    pairs = synthetic.make_pairs(sizes, 1000)
    g = graph.batch_pairs(pairs).to(dev); packed = g.pack()
    lig_w = torch.cat([torch.full((n, 1), 1.0 / (3 * n)) for n in packed.lig_counts]).to(dev)
    red = parallel.FlatGradAllReduce(net)
    T = {'zero': 0.0, 'fwd': 0.0, 'loss': 0.0, 'bwd': 0.0}
    def step():
        t0 = time.perf_counter(); red.zero()
        t1 = time.perf_counter(); o = net.forward_batched(g)
        t2 = time.perf_counter(); loss = bench.batched_loss(o[0], o[1], o[2], lig_w)
        t3 = time.perf_counter(); loss.backward()
        t4 = time.perf_counter()
        T['zero'] += t1 - t0; T['fwd'] += t2 - t1; T['loss'] += t3 - t2; T['bwd'] += t4 - t3
    for _ in range(10): step()
    torch.cuda.synchronize()
    for k in T: T[k] = 0.0
    for k in acc: acc[k] = 0.0
    n = 100
    for _ in range(n):
        step()
        if sizes[0][0] > 100: torch.cuda.synchronize()     # keep the queue empty: pure host cost, no back-pressure
    torch.cuda.synchronize()
    print(f"sizes {sizes[0]} x{len(sizes)} (us/step):", {k: round(v / n * 1e6) for k, v in T.items()}, {k: round(v / n * 1e6) for k, v in acc.items()})
