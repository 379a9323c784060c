"""Collate + layout + H->D of a batch: graph.batch_pairs (the role of the reference's
batchify_and_create_hetero_graphs + dgl.batch, src/utils/train_utils.py:61-108), PackedGraph.build (CSR / CSC / tiles /
attention work list / bf16 edge features) and the move to the GPU (one buffer per dtype, pinned staging).
usage (GPU box): python profiles/bench_collate.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from equidock_public_amd import graph as G, synthetic

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
# a DataLoader worker runs with ONE intra-op thread (torch sets that in every worker); with the main process' default
# (all cores of the box) the dozens of tiny host-side torch.cat calls of a collate spend their time in the thread pool
torch.set_num_threads(int(os.environ.get('EQD_COLLATE_THREADS', '1')))
print('intra-op threads', torch.get_num_threads())
for name, sizes in (('B: 8 x (200,200)', [(200, 200)] * 8), ('C: 64 x (300,300)', [(300, 300)] * 64)):
    pairs = synthetic.make_pairs(sizes, 1)
    n = 10
    t = {'batch_pairs': 0.0, 'pack (host)': 0.0, 'pin': 0.0, 'to(gpu)': 0.0}
    for it in range(n + 3):
        if it == 3:      # the first iterations pay one-time costs (pinned staging buffers, lazy imports)
            t = {k: 0.0 for k in t}
        t0 = time.perf_counter()
        g = G.batch_pairs(pairs)
        t1 = time.perf_counter()
        g.pack()
        t2 = time.perf_counter()
        g.pin_memory()                      # what DataLoader(pin_memory=True) does in its pin thread
        t2b = time.perf_counter()
        gd = g.to(dev)
        gd.pack()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        t['batch_pairs'] += t1 - t0; t['pack (host)'] += t2 - t1; t['pin'] += t2b - t2; t['to(gpu)'] += t3 - t2b
    print(name, {k: f'{v / n * 1e3:.2f} ms' for k, v in t.items()}, f"nodes {gd.pack().n_nodes} edges {gd.pack().n_edges}")
