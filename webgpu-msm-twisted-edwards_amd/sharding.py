"""Window sharding of one MSM over ranks (one process per GPU).

The W windows of the signed-digit decomposition are independent until Horner's rule
(submission.ts:369-407).  Rank r of D owns windows {w : w mod D == r}; every rank converts all n points
and decomposes only its own windows, reduces them on its GPU to W/D rows of 720 bytes, and the rows are
exchanged with ONE all-gather of W*720 bytes (11 KB for c = 16) -- RCCL over xGMI when the tensors are
on GPUs, gloo in the CPU tests.  Group addition is exact, so the result is bit-identical for every D.
"""
from __future__ import annotations

PARTIAL_BYTES = 720


def window_shard_for_rank(rank: int, world: int):
    """(first, step) such that the rank owns windows first, first+step, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank, world


def merge_partials(rows_per_rank, num_windows: int, world: int, row_bytes: int = PARTIAL_BYTES) -> bytes:
    """rows_per_rank[r] is rank r's full W x row_bytes buffer (only its own rows non-zero); row_bytes = 720, or 1120 for
    BLS12-377 G1."""
    out = bytearray(num_windows * row_bytes)
    for w in range(num_windows):
        r = w % world
        out[w * row_bytes:(w + 1) * row_bytes] = rows_per_rank[r][w * row_bytes:(w + 1) * row_bytes]
    return bytes(out)


def exchange_partials(partials, num_windows: int, dist=None, group=None, gather_list=None, row_bytes: int = PARTIAL_BYTES) -> bytes:
    """All-gathers every rank's W x 720 B tensor (CUDA tensor -> RCCL, CPU tensor -> gloo) and returns
    the merged rows as bytes.  `partials` must be complete on the current stream when called."""
    import torch

    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        if partials.is_cuda and dist.get_backend(group) == "gloo":
            partials, gather_list = partials.cpu(), None          # gloo has no CUDA all_gather; .cpu() synchronises the stream
        if gather_list is None:
            gather_list = [torch.empty_like(partials) for _ in range(world)]
        dist.all_gather(gather_list, partials, group=group)
        if partials.is_cuda:
            torch.cuda.current_stream().synchronize()
        return merge_partials([g.cpu().numpy().tobytes() for g in gather_list], num_windows, world, row_bytes)
    if partials.is_cuda:
        torch.cuda.current_stream().synchronize()
    return partials.cpu().numpy().tobytes()


def compute_msm_sharded(ctx, d_points, d_scalars, n: int, partials, dist=None, group=None, gather_list=None) -> bytes:
    """One window-sharded MSM on this rank's GPU.

    ctx        MsmContext whose window shard is (rank, world)
    d_points / d_scalars   torch uint8 CUDA tensors holding the FULL inputs on this rank's GPU
    partials   torch uint8 CUDA tensor of W * ctx.row_bytes bytes (720 per window, 1120 for BLS12-377 G1; scratch, overwritten)
    dist       torch.distributed (initialised) or None for a single rank
    Returns the 64-byte affine result (identical on every rank).  All device work is enqueued on
    torch's current stream so that the collective is ordered behind it.
    """
    import torch

    c, W = ctx.plan(n)
    partials.zero_()
    ctx.partial_device(d_points.data_ptr(), d_scalars.data_ptr(), n, partials.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    merged = exchange_partials(partials, W, dist, group, gather_list, ctx.row_bytes)
    return ctx.finalize(merged, c, W)


def distribute_inputs(points, scalars, dist=None, group=None, device="cuda", point_bytes: int = 64, scalar_bytes: int = 32):
    """Inputs for the window-sharded form arrive ONCE (SURVEY.md 8e "Inputs"; the reference uploads inside the call,
    cuzk/gpu.ts:33-46): every rank needs all n points and scalars, but rank r uploads only ITS n/D slice over its own PCIe
    link and one all-gather per buffer assembles the whole on every GPU -- RCCL over xGMI for CUDA tensors (12 MB per rank at
    n = 2^20, D = 8: each GPU receives 84 MB from seven peers), gloo for CPU tensors (tests, rehearsals).

    points / scalars   the full host buffers (bytes-like) as this rank's process holds them; only the rank's own slice
                       [rank * ceil(n / D), (rank + 1) * ceil(n / D)) of them is read and uploaded
    Returns (d_points, d_scalars, n): uint8 tensors on `device` holding the n points / scalars in order."""
    import warnings
    import torch

    n = len(scalars) // scalar_bytes
    if len(scalars) != n * scalar_bytes or len(points) != n * point_bytes:
        raise ValueError("distribute_inputs: points must be %d*n bytes and scalars %d*n bytes (got %d and %d)" % (point_bytes, scalar_bytes, len(points), len(scalars)))
    grouped = dist is not None and dist.is_initialized()         # (a group of one rank still runs the collective: the RCCL leg of the tests)
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    gloo = grouped and dist.get_backend(group) == "gloo"
    stage = "cpu" if gloo else device
    if grouped and world > 1:
        # every rank must bring the same n: a rank with another size would enter the all-gathers below with another buffer size
        # (a hang or garbage, depending on the backend) -- checked with a real exception on EVERY rank, before the collectives
        # that depend on it (asserts are stripped under -O: round-5 advisor)
        nn = torch.tensor([n, -n], dtype=torch.int64, device=stage)
        dist.all_reduce(nn, op=dist.ReduceOp.MAX, group=group)
        if int(nn[0].item()) != n or int(nn[1].item()) != -n:
            raise ValueError("distribute_inputs: the ranks disagree on n (this rank: %d, largest: %d, smallest: %d)" % (n, int(nn[0].item()), -int(nn[1].item())))
    per = (n + world - 1) // world                               # equal contributions: the last slice is padded
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)

    def mine(buf, unit):
        t = torch.zeros(per * unit, dtype=torch.uint8, device=stage)
        if hi > lo:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")              # (a read-only buffer: it is only read)
                src = torch.frombuffer(memoryview(buf)[lo * unit:hi * unit], dtype=torch.uint8)
            t[:(hi - lo) * unit].copy_(src, non_blocking=True)
        return t

    mp, ms = mine(points, point_bytes), mine(scalars, scalar_bytes)
    if not grouped:
        return mp[:n * point_bytes].to(device), ms[:n * scalar_bytes].to(device), n
    fp = torch.empty(world * per * point_bytes, dtype=torch.uint8, device=stage)
    fs = torch.empty(world * per * scalar_bytes, dtype=torch.uint8, device=stage)
    dist.all_gather_into_tensor(fp, mp, group=group)
    dist.all_gather_into_tensor(fs, ms, group=group)
    if fp.is_cuda:
        torch.cuda.current_stream().synchronize()
    return fp[:n * point_bytes].to(device), fs[:n * scalar_bytes].to(device), n


def rows_of_batched_msm(gathered, world: int, batch: int, m: int):
    """gathered: the all-gathered rows of a launch sequence that carried `batch` MSMs, a 1-D uint8 CPU tensor laid out
    [rank][MSM][W rows].  Returns MSM m's rows as a contiguous [rank][W rows] tensor -- the layout finalize_gathered takes."""
    return gathered.view(world, batch, -1)[:, m, :].contiguous()


class ShardedPipeline:
    """Window-sharded MSMs with `depth` launch sequences in flight per rank, each on its own stream and device work set: the
    all-gather, the read-back and the host tail of one overlap the device work of the following ones (the multi-GPU
    counterpart of te_msm_submit_device / te_msm_collect).  With batch > 1 a launch sequence carries up to `batch` MSMs
    (te_msm_partial_device_batch): a rank's W/D windows per MSM are too little work for kernels of their own, so the windows
    of several MSMs are sorted, accumulated and reduced together and their rows travel in one all-gather.

        pipe = ShardedPipeline(ctx, n, dist)          # ctx: MsmContext with window shard (rank, world)
        t = pipe.submit(d_points, d_scalars); ...; xy = pipe.collect(t)      # collect in submission order
        t = pipe.submit_batch([(p0, s0), (p1, s1)]); ...; [xy0, xy1] = pipe.collect_batch(t)
    """

    def __init__(self, ctx, n: int, dist, group=None, depth: int = 2, batch: int = 1):
        import torch
        from .binding import WORKSETS, MAX_BATCH

        assert 1 <= depth <= WORKSETS and 1 <= batch <= MAX_BATCH
        self.depth, self.batch = depth, batch

        self.torch, self.ctx, self.n, self.dist, self.group = torch, ctx, n, dist, group
        self.world = dist.get_world_size(group)
        self.c, self.W = ctx.plan(n)
        self.bucket_bits = self.c - 1 if ctx.get_option("signed_digits") else self.c
        self.curve = ctx.curve
        self.row_block = self.W * ctx.row_bytes                      # one MSM's rows
        nbytes = batch * self.row_block
        self.gloo = dist.get_backend(group) == "gloo"
        self.part = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(depth)]
        gdev = "cpu" if self.gloo else "cuda"
        self.gathered = [torch.zeros(self.world * nbytes, dtype=torch.uint8, device=gdev) for _ in range(depth)]
        self.host = [torch.zeros(self.world * nbytes, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream()
        # one stream per work set: the sequences overlap on the GPU.  torch's pool streams, left alone: both the context's own
        # streams (whose hardware queues the first te_msm_submit_device measures) through torch.cuda.ExternalStream and pool streams that had
        # been run through the same measurement were 8-15 % (four sequences in flight) to 70 % (eight) slower -- the runtime
        # binds a stream to a hardware queue when it is first used, and apparently binds better under load than an idle probe does.
        self.compute_streams = [torch.cuda.Stream() for _ in range(depth)]
        self.ev = [torch.cuda.Event() for _ in range(depth)]
        self.count = [0] * depth
        self.next_ticket = self.next_collect = 0
        self.inputs = None                                  # (d_points, d_scalars) once load_host() has distributed host buffers

    def load_host(self, points, scalars):
        """Distributes HOST buffers to the ranks' GPUs (distribute_inputs: rank r uploads its n / D slice, one all-gather per
        buffer over RCCL assembles the whole on every GPU) and keeps the result: (d_points, d_scalars) for submit()."""
        pb, sb = (96, 48) if self.curve == 1 else (64, 32)
        if len(scalars) != sb * self.n or len(points) != pb * self.n:      # before the collectives: a wrong size on one rank would desynchronise them
            raise ValueError("ShardedPipeline.load_host: the pipeline was planned for n = %d points (%d + %d bytes), got %d + %d bytes"
                             % (self.n, pb * self.n, sb * self.n, len(points), len(scalars)))
        dp, ds, n = distribute_inputs(points, scalars, self.dist, self.group, "cuda", pb, sb)
        self.inputs = (dp, ds)
        return self.inputs

    def submit(self, d_points=None, d_scalars=None) -> int:
        if d_points is None:
            if self.inputs is None:
                raise RuntimeError("ShardedPipeline.submit() without inputs: call load_host(points, scalars) first, or pass device tensors")
            d_points, d_scalars = self.inputs                   # what load_host left on this rank's GPU
        return self.submit_batch([(d_points, d_scalars)])

    def submit_batch(self, inputs) -> int:
        """inputs: 1..batch pairs (d_points, d_scalars) of CUDA uint8 tensors, all of n points."""
        torch = self.torch
        assert 1 <= len(inputs) <= self.batch, "between 1 and `batch` MSMs per launch sequence"
        assert self.next_ticket - self.next_collect < self.depth, "every slot has a launch sequence in flight"
        slot = self.next_ticket % self.depth
        self.count[slot] = len(inputs)
        cur = self.compute_streams[slot]
        cur.wait_stream(torch.cuda.current_stream())    # the caller's inputs are ready on its stream
        self.ctx.set_option("workset", slot)            # the slot's previous sequence was collected: its buffers are free
        with torch.cuda.stream(cur):                    # (rows of windows this rank does not own stay zero from allocation)
            if len(inputs) == 1:
                self.ctx.partial_device(inputs[0][0].data_ptr(), inputs[0][1].data_ptr(), self.n, self.part[slot].data_ptr(), cur.cuda_stream)
            else:
                self.ctx.partial_device_batch([p.data_ptr() for p, _ in inputs], [s.data_ptr() for _, s in inputs], self.n,
                                              self.part[slot].data_ptr(), cur.cuda_stream)
            if self.gloo:                               # rehearsal path (no CUDA all_gather in gloo): blocking
                src = self.part[slot].cpu()
                self.dist.all_gather_into_tensor(self.gathered[slot], src, group=self.group)
                self.host[slot].copy_(self.gathered[slot])
                self.ev[slot].record(cur)
            else:                                       # the collective is ordered behind this slot's compute stream
                work = self.dist.all_gather_into_tensor(self.gathered[slot], self.part[slot], group=self.group, async_op=True)
        if not self.gloo:
            with torch.cuda.stream(self.copy_stream):   # the copy stream, not a compute stream, waits for the collective
                work.wait()
                self.host[slot].copy_(self.gathered[slot], non_blocking=True)
                self.ev[slot].record(self.copy_stream)
        t = self.next_ticket
        self.next_ticket += 1
        return t

    def collect(self, ticket: int) -> bytes:
        out = self.collect_batch(ticket)
        assert len(out) == 1, "a batch was submitted under this ticket: use collect_batch"
        return out[0]

    def collect_batch(self, ticket: int):
        from .binding import finalize_gathered
        assert ticket == self.next_collect, "collect in submission order"
        slot = ticket % self.depth
        self.ev[slot].synchronize()
        self.next_collect += 1
        self.ctx.partial_wait(slot)                     # done already (the copy is ordered behind it): reports scalar-range errors
        if self.batch == 1:
            return [finalize_gathered(self.host[slot].data_ptr(), self.world, self.c, self.W, self.bucket_bits, self.curve)]
        out = []
        for m in range(self.count[slot]):
            mine = rows_of_batched_msm(self.host[slot], self.world, self.batch, m)
            out.append(finalize_gathered(mine.data_ptr(), self.world, self.c, self.W, self.bucket_bits, self.curve))
        return out
