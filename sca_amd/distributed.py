"""Agents sharded over the GPUs of one node: one process per GPU, contiguous id ranges, ONE exchange per step.

Per step and rank:
    step_begin : kd-tree (replicated) -> neighbours -> solve -> integrate, for the rank's shard only
    all-gather : the shard's moved 48-byte public records (RCCL over xGMI; torch.distributed backend "nccl")
    step_end   : collision flags for the shard (needs everyone's moved records), at-goal flags for everyone, publish
Every rank holds all N public records; private per-agent state is only meaningful for the owner's shard.

Two ways to run the exchange:
  * inlib=True  -- RCCL inside the library (sca_comm_init): `run(k)` is ONE sca_run_steps call, the ncclAllGather of every
    step is enqueued by the library on its own stream between its kernels.  What bench.py --gpus N uses.
  * otherwise   -- the collective is issued from here (torch.distributed) between sca_step_begin and sca_step_end: the path
    for callers that already own a process group / their own buffers, and the test double of the first.
The backend only has to offer set_shard / step_begin / step_end / run_steps / synchronize plus `moved_records()`
returning (full tensor, this rank's slice) -- tests drive the same class on CPU with gloo and a checker backend.
"""


class ShardedStepper:
    def __init__(self, solver, rank=0, world=1, torch_mod=None, dist_mod=None, mode=0, staged=False, force_exchange=False,
                 inlib=False, unique_id=None):
        """staged=True exchanges through host memory (for backends without GPU collectives, e.g. gloo when several
        ranks share one GPU in tests); the default hands the device buffers to the collective directly (RCCL).
        inlib=True: the library's own RCCL communicator (unique_id: the 128 bytes of rank 0's comm_unique_id())."""
        self.staged = staged
        self.inlib = bool(inlib)
        self.exchange = int(world) > 1 or force_exchange      # force_exchange: the begin / all-gather / end path even alone
        self.sol = solver
        self.rank, self.world = int(rank), int(world)
        self.torch = torch_mod
        self.dist = dist_mod
        self.mode = mode
        self._stream = None
        self.measure_exchange = False           # True: an event pair around every collective issued from here (exchange_ms())
        self._xch_events = []
        n = solver.n
        if self.exchange:
            if n % self.world:
                raise ValueError(f'{n} agents do not split evenly over {self.world} ranks')
            self.count = n // self.world
            self.begin = self.rank * self.count
            if self.inlib:
                solver.comm_init(self.rank, self.world, unique_id)      # sets the shard
            else:
                solver.set_shard(self.begin, self.count)
                self._setup_exchange()
        else:
            self.begin, self.count = 0, n

    def _setup_exchange(self):
        sol = self.sol
        if hasattr(sol, 'moved_records'):
            return
        torch = self.torch
        nbytes = sol.public_records(1)[1] * sol.n
        dev = torch.device('cuda', torch.cuda.current_device())
        self._cur = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self._moved = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        sol.bind_public_records(self._cur.data_ptr(), self._moved.data_ptr(), nbytes)
        # The library's kernels and the collective must be ordered on ONE stream: the stepper owns a torch stream, the
        # library launches on it, and the collective is issued with it current (ProcessGroupNCCL waits for / is waited on
        # by the current stream).  (The library's default is a non-blocking stream of its own, which nothing in torch
        # would order against.)
        torch.cuda.synchronize()                      # the two buffers were zeroed on the default stream
        self._stream = torch.cuda.Stream(device=dev)
        sol.set_stream(self._stream.cuda_stream)
        self._per = nbytes // self.world
        self._by_ptr = {self._cur.data_ptr(): self._cur, self._moved.data_ptr(): self._moved}

    def _moved_records(self):
        """(full buffer, this rank's slice) of the records sca_step_begin has just written.  The library swaps its two
        record buffers at the end of every step, so the buffer is looked up by address."""
        if hasattr(self.sol, 'moved_records'):
            return self.sol.moved_records()
        full = self._by_ptr[self.sol.public_records(1)[0]]
        return full, full[self.rank * self._per:(self.rank + 1) * self._per]

    def run(self, steps):
        if not self.exchange or self.inlib:
            self.sol.run_steps(steps, self.mode)                     # with a communicator: the exchange is inside
            return
        for _ in range(int(steps)):
            self.sol.step_begin(self.mode)
            full, mine = self._moved_records()
            if self.staged:
                self.sol.synchronize()
                h_mine = mine.cpu()
                h_full = self.torch.empty(full.numel(), dtype=full.dtype)
                self.dist.all_gather_into_tensor(h_full, h_mine)
                full.copy_(h_full)
                self.torch.cuda.synchronize()
            elif getattr(self, '_stream', None) is not None:
                with self.torch.cuda.stream(self._stream):           # ordered with the library's kernels (see _setup_exchange)
                    if self.measure_exchange:
                        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                        e0.record(self._stream)
                        self.dist.all_gather_into_tensor(full, mine)
                        e1.record(self._stream)
                        self._xch_events.append((e0, e1))
                    else:
                        self.dist.all_gather_into_tensor(full, mine)
            else:                                                    # a backend with its own buffers (the CPU checker of the tests)
                self.dist.all_gather_into_tensor(full, mine)
            self.sol.step_end()

    def sync(self):
        self.sol.synchronize()

    def exchange_ms(self, reset=True):
        """Mean device time of the step's collective over the steps run with measure_exchange (torch path: events on the stepper's
        stream around all_gather_into_tensor; inlib: the library's own event pair around ncclAllGather, needs set_profiling).  None
        when nothing was measured (one rank, staged test exchange)."""
        if self.inlib:
            v = self.sol.exchange_ms()
            return v if v > 0 else None
        if not self._xch_events:
            return None
        self.sol.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._xch_events]
        if reset:
            self._xch_events = []
        return sum(ms) / len(ms)


class PartitionedStepper:
    """SCA_NBR_GRID with the cell-owner partition (sca_partition_*): slabs of grid cells along one axis, one
    per rank; a step exchanges, with the two slab neighbours only, the records of the agents next to the cut and the agents
    that crossed it -- point to point (isend / irecv: RCCL on GPUs; gloo with host staging when ranks share a GPU in tests).

    Every rank must hold the complete state when this is constructed (sca_set_state); afterwards a rank's per-agent arrays are
    meaningful for `owned()` only."""

    def __init__(self, solver, rank, world, torch_mod, dist_mod, axis=0, cuts=None, staged=False, cap_halo=0, cap_mig=0, emulate=False):
        """emulate=True (measurements on one GPU): this rank of `world` alone -- its messages go nowhere and empty ones arrive, so
        the halo copies are gone after the first step and agents that leave are lost; what the rank executes per step is what it
        would execute in company."""
        self.emulate = bool(emulate)
        if self.emulate:
            solver.set_shard_emulation(True)
        self.sol, self.rank, self.world = solver, int(rank), int(world)
        self.torch, self.dist, self.staged = torch_mod, dist_mod, staged
        solver.partition_init(self.rank, self.world, axis, cuts, cap_halo, cap_mig)
        nbytes = solver.partition_message_bytes()
        dev = torch_mod.device('cuda', torch_mod.cuda.current_device())
        self.out = [torch_mod.zeros(nbytes, dtype=torch_mod.uint8, device=dev) for _ in range(2)]     # to the lower / upper neighbour
        self.inb = [torch_mod.zeros(nbytes, dtype=torch_mod.uint8, device=dev) for _ in range(2)]     # from the lower / upper neighbour
        self._stream = None
        if not staged and self.world > 1 and not self.emulate:
            torch_mod.cuda.synchronize()
            self._stream = torch_mod.cuda.Stream(device=dev)      # the library's kernels and the transfers on ONE stream
            solver.set_stream(self._stream.cuda_stream)

    def owned(self):
        return self.sol.partition_owned()

    def _peers(self):
        return [(0, self.rank - 1) if self.rank > 0 else None, (1, self.rank + 1) if self.rank + 1 < self.world else None]

    def _exchange(self):
        t, d = self.torch, self.dist
        if self.world == 1 or self.emulate:                      # (the inbound buffers stay zeroed: empty messages)
            return
        if self.staged:
            self.sol.synchronize()
            send = [self.out[s].cpu() for s in (0, 1)]
            recv = [t.empty_like(send[0]) for _ in (0, 1)]
            ops = []
            for pr in self._peers():
                if pr is None:
                    continue
                side, peer = pr
                ops.append(d.P2POp(d.isend, send[side], peer))
                ops.append(d.P2POp(d.irecv, recv[side], peer))
            for w in d.batch_isend_irecv(ops):
                w.wait()
            for pr in self._peers():
                if pr is not None:
                    self.inb[pr[0]].copy_(recv[pr[0]])
            t.cuda.synchronize()
            return
        with t.cuda.stream(self._stream):
            ops = []
            for pr in self._peers():
                if pr is None:
                    continue
                side, peer = pr
                ops.append(d.P2POp(d.isend, self.out[side], peer))
                ops.append(d.P2POp(d.irecv, self.inb[side], peer))
            for w in d.batch_isend_irecv(ops):
                w.wait()

    def run(self, steps):
        from .solver import NBR_GRID
        sol = self.sol
        if self.world == 1 or self.emulate:                      # nobody to talk to: the whole loop inside the library
            sol.run_steps(steps, NBR_GRID)
            return
        for _ in range(int(steps)):
            sol.step_begin(NBR_GRID)
            sol.partition_pack(self.out[0].data_ptr(), self.out[1].data_ptr())
            self._exchange()
            sol.partition_unpack(self.inb[0].data_ptr(), self.inb[1].data_ptr())
            sol.partition_commit()
            sol.step_end()

    def sync(self):
        self.sol.synchronize()
