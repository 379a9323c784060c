"""The training step of the reference for a PairGraph batch, as the library runs it (src/train.py:98-154):

    model(batch) -> per pair: MSE(ligand) + pocket OT (exact EMD) + body intersection -> batch means, weighted sum -> backward

The reference walks the pairs in a Python loop and calls POT's exact solver on the host once per pair, with a device -> host
synchronisation each (src/utils/ot_utils.py:22-29).  Here every term is batched on the device (eqd_pair_losses_*,
eqd_pocket_ot_*; losses.py) and the exact transport plans - tiny, branchy problems that belong on the host - are solved by
libequidock_host.so on worker threads with ONE device <-> host round trip per step.  `TrainStep` arranges the step so that
this round trip is the only host join and everything around it replays from hipGraphs:

    graph F   zero-grad fill, eqd_model_forward, cost matrices of all pairs (eqd_pocket_ot_cost), async D -> H into pinned memory
    graph M   MSE + intersection forward and their gradient w.r.t. the ligand coordinates (needs no plan: the GPU works on it
              while the host waits for the cost matrices and solves)
    host      wait for the cost copy, solve B transport problems, async H -> D of the plans
    graph B   sum(plan * cost), its gradient w.r.t. the keypoints, eqd_model_backward (+ the RCCL all-reduce of the flat gradient)

`step()` returns the loss tensor; `last_ot_exposed_ms()` is the stretch of the GPU timeline between the end of graph M and the
start of graph B - what the host solve and its two copies cost the step.  `step_eager()` is the same arithmetic through the
autograd wrappers of losses.py (what tests compare with the oracle).  No CPU fallback: tensors must be on the GPU.
"""
import ctypes as C

import torch

from . import _lib, losses


class TrainStep:

    def __init__(self, net, batch, lig_target, rec, pocket_lig_list, pocket_rec_list, w_ot=1.0, w_int=10.0, sigma=25.0,
                 surface_ct=10.0, reducer=None, n_threads=0, allreduce=False):
        """lig_target [n_lig, 3] / rec [n_rec, 3]: bound coordinates in the batch's node order (src/train.py:114, 131);
        pocket_*_list: per pair (n_pocket, 3), matched rows; weights / sigma / surface_ct: src/utils/args.py:64-70;
        reducer: parallel.FlatGradAllReduce of `net` (gradients accumulate in its flat buffer); allreduce: issue the
        collective at the end of the backward (a process group must exist)."""
        from .parallel import FlatGradAllReduce
        self.net, self.batch = net, batch
        self.packed = batch.pack()
        dev = self.packed.x0.device
        self.dev = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.reducer = reducer if reducer is not None else FlatGradAllReduce(net)
        self.allreduce = bool(allreduce)
        self.w_ot, self.w_int, self.sigma, self.ct = float(w_ot), float(w_int), float(sigma), float(surface_ct)
        self.n_threads = int(n_threads)
        self.lig_target = _lib.require_device(lig_target.to(dev, torch.float32).contiguous(), 'lig_target')
        self.rec = _lib.require_device(rec.to(dev, torch.float32).contiguous(), 'rec')
        self.pl_list = [t.to(dev) for t in pocket_lig_list]
        self.pr_list = [t.to(dev) for t in pocket_rec_list]
        counts = [int(t.shape[0]) for t in pocket_lig_list]
        if counts != [int(t.shape[0]) for t in pocket_rec_list] or len(counts) != self.packed.n_pairs:
            raise ValueError("pocket lists must have one (n_pocket, 3) tensor per pair, same rows for ligand and receptor")
        self.counts = counts
        self.B, self.K = self.packed.n_pairs, int(net.iegmn_original.num_att_heads)
        B, K, rows = self.B, self.K, sum(counts)
        self.pl = torch.cat([t.reshape(-1, 3) for t in self.pl_list]).to(**f32).contiguous()
        self.pr = torch.cat([t.reshape(-1, 3) for t in self.pr_list]).to(**f32).contiguous()
        self.off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32).to(dev)
        self.counts_host = torch.tensor(counts, dtype=torch.int32)
        # static buffers (the graphs address them)
        self.cost = torch.empty(rows, K, **f32)
        self.plan = torch.empty(rows, K, **f32)
        pin = dev.type == 'cuda'      # (the simulator of the tests runs this class on host tensors: no pinning, no events)
        self.cost_host = torch.empty(rows, K, dtype=torch.float32, pin_memory=pin)
        self.plan_host = torch.empty(rows, K, dtype=torch.float32, pin_memory=pin)
        self.values_host = torch.empty(B, dtype=torch.float64)
        self.mse, self.inter, self.ot = torch.empty(B, **f32), torch.empty(B, **f32), torch.zeros(B, **f32)
        self.s_lig, self.s_rec = torch.empty(self.packed.n_lig, **f32), torch.empty(self.packed.n_rec, **f32)
        self.d_lig = torch.empty(self.packed.n_lig, 3, **f32)
        self.dYl, self.dYr = torch.empty(B, K, 3, **f32), torch.empty(B, K, 3, **f32)
        self.d_mse = torch.full((B,), 1.0 / B, **f32)                  # d loss / d mse[p]   (src/train.py:143-150: batch means)
        self.d_inter = torch.full((B,), self.w_int / B, **f32)
        self.d_ot = torch.full((B,), self.w_ot / B, **f32)
        self.loss = torch.zeros((), **f32)
        self._graphs = None
        self._outs = None
        self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if dev.type == 'cuda' else None
        self._exposed = []

    # ---- the pieces (each enqueues on the current stream; none synchronises) -------------------------------------------
    def _forward_and_cost(self):
        lib = _lib.load_library()
        self.reducer.zero()
        lig, Yl, Yr, T, b = self.net.forward_batched(self.batch)
        self._outs = (lig, Yl, Yr)
        with _lib.device_guard(self.dev):
            _lib.check(lib.eqd_pocket_ot_cost(self.B, self.K, _lib.ptr(self.off), _lib.ptr(self.pl), _lib.ptr(self.pr),
                                              _lib.ptr(Yl.detach()), _lib.ptr(Yr.detach()), _lib.ptr(self.cost),
                                              _lib.stream_ptr(self.dev)))
        self.cost_host.copy_(self.cost, non_blocking=True)

    def _pair_terms(self):
        lib = _lib.load_library()
        lig = self._outs[0].detach()
        gs = self.packed.c_struct()
        st = _lib.stream_ptr(self.dev)
        with _lib.device_guard(self.dev):
            _lib.check(lib.eqd_pair_losses_fwd(C.byref(gs), _lib.ptr(lig), _lib.ptr(self.lig_target), _lib.ptr(self.rec),
                                               C.c_float(self.sigma), C.c_float(self.ct), _lib.ptr(self.mse), _lib.ptr(self.inter),
                                               _lib.ptr(self.s_lig), _lib.ptr(self.s_rec), st))
            _lib.check(lib.eqd_pair_losses_bwd(C.byref(gs), _lib.ptr(lig), _lib.ptr(self.lig_target), _lib.ptr(self.rec),
                                               C.c_float(self.sigma), C.c_float(self.ct), _lib.ptr(self.s_lig), _lib.ptr(self.s_rec),
                                               _lib.ptr(self.d_mse), _lib.ptr(self.d_inter), _lib.ptr(self.d_lig), st))

    def _solve_on_host(self):
        """exact plans of the B transport problems, pinned buffer to pinned buffer (the caller has waited for the cost copy)"""
        lib = losses._emd_lib()
        rc = lib.eqd_host_emd_uniform(self.B, C.c_void_p(self.counts_host.data_ptr()), self.K,
                                      C.c_void_p(self.cost_host.data_ptr()), C.c_void_p(self.plan_host.data_ptr()),
                                      C.c_void_p(self.values_host.data_ptr()), self.n_threads)
        if rc != 0:
            raise _lib.EquidockHipError("exact transport solver failed (non-finite cost matrix?)")

    def _ot_and_backward(self):
        lib = _lib.load_library()
        lig, Yl, Yr = self._outs
        st = _lib.stream_ptr(self.dev)
        with _lib.device_guard(self.dev):
            _lib.check(lib.eqd_pocket_ot_fwd(self.B, self.K, _lib.ptr(self.off), _lib.ptr(self.plan), _lib.ptr(self.cost),
                                             _lib.ptr(self.ot), st))
            _lib.check(lib.eqd_pocket_ot_bwd(self.B, self.K, _lib.ptr(self.off), _lib.ptr(self.pl), _lib.ptr(self.pr),
                                             _lib.ptr(Yl.detach()), _lib.ptr(Yr.detach()), _lib.ptr(self.plan), _lib.ptr(self.d_ot),
                                             _lib.ptr(self.dYl), _lib.ptr(self.dYr), st))
        # loss = mean(mse) + w_ot mean(ot) + w_int mean(inter)   (src/train.py:143-150)
        torch.add(self.mse.mean() + self.w_ot * self.ot.mean(), self.inter.mean(), alpha=self.w_int, out=self.loss)
        torch.autograd.backward([lig, Yl, Yr], [self.d_lig, self.dYl, self.dYr])
        self._outs = None
        if self.allreduce:
            self.reducer.reduce(force=True)

    # ---- the step ------------------------------------------------------------------------------------------------------
    def _join(self, fwd, mid, bwd):
        """fwd, mid, bwd: callables that enqueue the three device parts (graph replays or eager launches)"""
        if self.dev.type != 'cuda':                # (simulator: everything is synchronous)
            fwd()
            mid()
            self._solve_on_host()
            self.plan.copy_(self.plan_host)
            bwd()
            return self.loss
        cur = torch.cuda.current_stream(self.dev)
        copied = torch.cuda.Event()
        fwd()
        copied.record(cur)
        mid()
        self._ev[0].record(cur)
        copied.synchronize()                       # the ONE host join of the step: the cost matrices are in pinned memory
        self._solve_on_host()
        self.plan.copy_(self.plan_host, non_blocking=True)
        self._ev[1].record(cur)
        bwd()
        return self.loss

    def step_unfused(self):
        """the same step with every launch enqueued from the host (no graphs)"""
        return self._join(self._forward_and_cost, self._pair_terms, self._ot_and_backward)

    def capture(self):
        """capture the three device parts into hipGraphs that share one memory pool (the forward's saved state lives in it
        until the backward graph has consumed it)"""
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.step_unfused()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gf, gm, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, capture_error_mode='thread_local'):
            self._forward_and_cost()
        pool = gf.pool()
        with torch.cuda.graph(gm, pool=pool, capture_error_mode='thread_local'):
            self._pair_terms()
        self.plan.copy_(self.plan_host, non_blocking=True)      # (a plan of the warm-up steps: only its shape matters here)
        with torch.cuda.graph(gb, pool=pool, capture_error_mode='thread_local'):
            self._ot_and_backward()
        self._graphs = (gf, gm, gb)
        torch.cuda.synchronize()
        return self

    def step(self):
        if self._graphs is None:
            return self.step_unfused()
        gf, gm, gb = self._graphs
        return self._join(gf.replay, gm.replay, gb.replay)

    def last_ot_exposed_ms(self):
        """GPU-timeline gap between the end of graph M and the start of graph B of the last step (synchronises)"""
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])

    def step_eager(self):
        """the same arithmetic through the autograd wrappers of losses.py (pair_losses, pocket_ot_loss): one blocking
        round trip inside pocket_ot_loss, no graphs - the form tests compare with the oracle"""
        self.reducer.zero()
        lig, Yl, Yr, T, b = self.net.forward_batched(self.batch)
        mse, inter = losses.pair_losses(self.batch, lig, self.lig_target, self.rec, self.sigma, self.ct)
        ot = losses.pocket_ot_loss(Yl, Yr, self.pl_list, self.pr_list, n_threads=self.n_threads)
        loss = mse.mean() + self.w_ot * ot.mean() + self.w_int * inter.mean()
        loss.backward()
        if self.allreduce:
            self.reducer.reduce(force=True)
        return loss
