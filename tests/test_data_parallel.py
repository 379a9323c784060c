"""Data-parallel path on CPU: world_size 2 over gloo, compute through the x86 simulator of the C ABI.
N-rank averaged gradients must equal the single-process gradient of the concatenated batch
(SURVEY.md section 4 (4)); shard_pairs must be a balanced partition."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [(30, 41), (52, 27), (25, 33), (44, 38)]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=9):
    from equidock_public_amd import model as M
    from oracle import iegmn_port as port
    args = port.default_args(iegmn_n_lays=2, skip_weight_h=0.5)
    net = M.Rigid_Body_Docking_Net(args)
    net.load_state_dict(port.init_state_dict(args, seed))
    return net


def _loss(net, g):
    lig, Yl, Yr, T, b = net.forward_batched(g)
    # mean over the local pairs of a per-pair loss (equal local batch sizes -> mean of rank means = global mean)
    return (lig.square().sum() / 100.0 + Yl.square().mean() + Yr.square().mean() + (T * T).mean() + b.square().mean())


def _worker(rank, world, port_no, lib, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port_no)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from equidock_public_amd import _lib, graph as G, parallel, synthetic
    _lib.load_library_for_testing(lib)
    net = _make(seed=9 + rank)                 # different init per rank: broadcast must fix it
    parallel.broadcast_parameters(net, src=0)
    pairs = synthetic.make_pairs(SIZES, 77)
    mine = parallel.shard_pairs(SIZES, world, rank)
    g = G.batch_pairs([pairs[i] for i in mine])
    red = parallel.FlatGradAllReduce(net)
    red.zero()
    # local loss: sum over local pairs of lig^2/100 ... normalised so that the rank average equals the
    # global-batch loss below
    lig, Yl, Yr, T, b = net.forward_batched(g)
    loss = lig.square().sum() / 100.0 * world + Yl.square().mean() + Yr.square().mean() + (T * T).mean() + b.square().mean()
    loss.backward()
    flat = red.reduce()
    torch.save({'flat': flat.clone(), 'mine': mine}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_pairs_is_balanced_partition():
    from equidock_public_amd import parallel
    sizes = [(10 * i + 5, 300 - 7 * i) for i in range(13)]
    for world in (1, 2, 4, 8):
        shards = [parallel.shard_pairs(sizes, world, r) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(len(sizes)))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_two_rank_gradients_equal_single_process(tmp_path):
    from tests.hostsim import build as hs
    lib = hs.build()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), lib, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    assert torch.equal(r0['flat'], r1['flat'])            # every rank holds the same averaged gradient
    assert sorted(r0['mine'] + r1['mine']) == [0, 1, 2, 3]
    # single process, whole batch (pairs in shard order), same per-pair loss
    from equidock_public_amd import _lib, graph as G, synthetic
    _lib.load_library_for_testing(lib)
    try:
        net = _make(seed=9)
        pairs = synthetic.make_pairs(SIZES, 77)
        order = r0['mine'] + r1['mine']
        g = G.batch_pairs([pairs[i] for i in order])
        flat = net.iegmn_original.enable_flat_grads()
        lig, Yl, Yr, T, b = net.forward_batched(g)
        loss = lig.square().sum() / 100.0 + Yl.square().mean() + Yr.square().mean() + (T * T).mean() + b.square().mean()
        loss.backward()
    finally:
        _lib.unload_for_testing()
    ref = flat
    err = float((r0['flat'] - ref).norm() / ref.norm())
    assert err < 1e-5, err


def test_bench_gpus2_relaunches_itself_under_torchrun():
    """`python bench.py --gpus 2` WITHOUT a launcher (no RANK in the environment) must re-execute itself under
    torch.distributed.run with two ranks, print exactly one JSON line (rank 0) and propagate the exit code.  Here (no GPU)
    the ranks run the launcher rehearsal (EQD_BENCH_DRY_RUN=1: gloo rendezvous, barrier, max over ranks); the same
    command with the workload runs on the GPU box in tests/test_gpu_parity.py."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['EQD_BENCH_DRY_RUN'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['dry_run'] is True and out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1
    assert 'torch.distributed.run' in r.stderr
    # a failing rank must surface as a non-zero exit code of the plain command
    env['EQD_BENCH_DRY_RUN_FAIL'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
