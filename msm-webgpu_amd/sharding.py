"""Window sharding across the GPUs of one node (new relative to the reference, which is single-device).

Pippenger windows are independent: rank r computes the window sums S_w for its contiguous
window range on its own GPU (all bases resident on every GPU, every rank recodes all scalars because the signed-digit
carry chain runs across windows), then ONE collective -- an all-gather of world_size x W_local x 96 B over RCCL/xGMI --
brings all 16 window sums to every rank (a record is 96 B on BN254 G1; the context's curve decides: ctx.jb), and the host window combine (src/cuzk/msm.rs:411-416) finishes.  Elliptic-curve
addition is not an RCCL reduction operator, hence gather + local combine rather than all-reduce.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .api import NUM_WINDOWS, MsmContext


def window_range(rank, world_size, num_windows=NUM_WINDOWS):
    """Contiguous, balanced partition of [0, num_windows): the first (num_windows % world) ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(num_windows, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_windows_per_rank(world_size, num_windows=NUM_WINDOWS):
    return -(-num_windows // world_size)


def gather_window_sums(local_sums, rank, world_size, group=None, num_windows=NUM_WINDOWS, jb=96):
    """local_sums: uint8 [W_local, jb] (device tensor under nccl, CPU tensor under gloo); jb = bytes of a Jacobian record of the curve
    (96; 144 BLS12-381; 192 / 288 the G2 curves).  Returns uint8 [num_windows, jb] with every window sum in window order, on every rank."""
    per = max_windows_per_rank(world_size, num_windows)
    padded = torch.zeros((per, jb), dtype=torch.uint8, device=local_sums.device)
    padded[: local_sums.shape[0]] = local_sums
    if world_size == 1:
        gathered = padded.unsqueeze(0)
    else:
        gathered = torch.empty((world_size, per, jb), dtype=torch.uint8, device=local_sums.device)
        dist.all_gather_into_tensor(gathered.view(-1), padded.view(-1), group=group)
    rows = []
    for r in range(world_size):
        b, e = window_range(r, world_size, num_windows)
        rows.append(gathered[r, : e - b])
    return torch.cat(rows, dim=0)


def sharded_msm(ctx, scalars_dev, rank, world_size, group=None):
    """One MSM with windows sharded over the ranks of `group`; every rank returns the full G1 result."""
    b, e = window_range(rank, world_size)
    if e > b:
        local = ctx.msm_windows(scalars_dev, b, e)
    else:
        local = torch.empty((0, ctx.jb), dtype=torch.uint8, device=scalars_dev.device)
    all_sums = gather_window_sums(local, rank, world_size, group, jb=ctx.jb)
    return MsmContext.combine_windows(all_sums, curve=ctx.curve)


def batch_range(rank, world_size, batch):
    """Contiguous, balanced share of `batch` independent MSMs for `rank` (BASELINE config 5: many MSMs, one shared base)."""
    return window_range(rank, world_size, batch)


def gather_batch_results(local_results, rank, world_size, batch, group=None, jb=96):
    """local_results: uint8 [B_local, jb] Jacobian records of this rank's share (device tensor under nccl, CPU tensor
    under gloo).  Returns uint8 [batch, jb] in MSM order on every rank -- the only collective of the batch-sharded path."""
    per = max_windows_per_rank(world_size, batch)
    padded = torch.zeros((per, jb), dtype=torch.uint8, device=local_results.device)
    padded[: local_results.shape[0]] = local_results
    if world_size == 1:
        gathered = padded.unsqueeze(0)
    else:
        gathered = torch.empty((world_size, per, jb), dtype=torch.uint8, device=local_results.device)
        dist.all_gather_into_tensor(gathered.view(-1), padded.view(-1), group=group)
    rows = []
    for r in range(world_size):
        b, e = batch_range(r, world_size, batch)
        rows.append(gathered[r, : e - b])
    return torch.cat(rows, dim=0)


def sharded_batch_msm(ctx, scalars_dev, n, rank, world_size, group=None):
    """`batch` independent MSMs over the resident bases, whole MSMs sharded over the ranks (no exchange on the data path;
    one all-gather of batch x 96 B at the end).  scalars_dev: CUDA uint8 [batch * n, 32], identical on every rank (or at
    least this rank's share valid).  Returns the list of all G1 results on every rank."""
    from .api import G1

    batch = scalars_dev.shape[0] // n
    b, e = batch_range(rank, world_size, batch)
    mine = ctx.msm_batch(scalars_dev[b * n:e * n], n) if e > b else []
    dev = scalars_dev.device
    jb = ctx.jb
    local = torch.tensor(list(b"".join(g.xyz for g in mine)), dtype=torch.uint8, device=dev).view(len(mine), jb)
    allr = gather_batch_results(local, rank, world_size, batch, group, jb=jb).cpu().numpy().tobytes()
    return [G1(allr[jb * k:jb * k + jb], ctx.modulus) for k in range(batch)]


def msms_per_launch(world_size, num_windows=NUM_WINDOWS):
    """How many MSMs a rank processes per launch in the window-sharded pipeline: as many as fit 16 local windows.  A rank's
    share of ONE MSM (2 windows at 8 GPUs) is too small to fill a GPU -- kernel latencies, not work, set its time -- so the
    shares of several independent MSMs go through one kernel sequence (msm_hip_launch_windows_batch_device)."""
    return max(1, num_windows // max_windows_per_rank(world_size, num_windows))


def group_window_rows(gathered, v, world_size, num_windows=NUM_WINDOWS, window_ranges=None):
    """gathered: uint8 [world, rows, 96] where rank r wrote its records vector-major ([nvec][w_r][96], w_r = its window count)
    at the front of its row block.  Returns the num_windows x 96 records of vector `v` in window order."""
    rows = []
    for r in range(world_size):
        b, e = window_ranges[r] if window_ranges else window_range(r, world_size, num_windows)
        rows.append(gathered[r, v * (e - b):(v + 1) * (e - b)])
    return torch.cat(rows, dim=0)


def gathered_window_sums(host, nvec, world_size, num_windows=NUM_WINDOWS):
    """host: numpy uint8 [world, rows, jb], the all-gathered blocks of one launch (rank r's block holds [nvec][its windows] records,
    vector-major, then padding).  Returns a contiguous uint8 array [nvec, num_windows, jb]: every MSM's window sums in window order --
    the input of ONE msm_hip_combine_windows_batch_curve call for the whole launch."""
    jb = host.shape[-1]
    if num_windows % world_size == 0:  # equal shares: one transpose
        per = num_windows // world_size
        return np.ascontiguousarray(host[:, : nvec * per].reshape(world_size, nvec, per, jb).transpose(1, 0, 2, 3)).reshape(nvec, num_windows, jb)
    out = np.empty((nvec, num_windows, jb), dtype=np.uint8)
    for r in range(world_size):
        b, e = window_range(r, world_size, num_windows)
        out[:, b:e] = host[r, : nvec * (e - b)].reshape(nvec, e - b, jb)
    return out


class ShardedMsmPipeline:
    """Back-to-back window-sharded MSMs with everything asynchronous: rank-local device work (result slots and
    main/reduce HIP streams inside the engine), the RCCL all-gather on the torch stream (ordered after the slot by a device-side
    event wait, no host sync), the D2H copy into pinned memory, and the host window combine one step behind.

        pipe = ShardedMsmPipeline(ctx, rank, world_size, group)
        pipe.issue(scalars_0); pipe.issue(scalars_1); r0 = pipe.complete(); pipe.issue(scalars_2); r1 = pipe.complete(); ...
    At most `depth` (<= 3) launches may be in flight; each uses one of the engine's four result slots.

    `group` > 1: every issue() takes `group` (or fewer) scalar vectors at once (CUDA uint8 [g * n, 32], contiguous) and
    complete() returns the list of their results; one launch and ONE all-gather serve all of them.
    """

    SLOTS = 4

    def __init__(self, ctx, rank, world_size, group=None, num_windows=None, depth=3, msms_per_issue=1, emulate_world=0, halves=False,
                 combine="all", wide=False):
        """halves: the context's bases carry their endomorphism images (set_bases(..., endomorphism=True)); the ranks then share the 8
        HALF-length windows of the 2n-point problem (msm_hip_launch_half_windows_batch_device) instead of the 16 full-length ones.
        combine: who runs the host window combine (src/cuzk/msm.rs:411-416) of a launch's MSMs -- every rank holds all window sums after
        the all-gather.  "all": every rank combines every MSM (every rank returns every result; world x the host work).  "spread": vector v
        of a launch is combined ONCE, by rank v % world -- complete() returns None in the other ranks' places (the throughput form: each
        result exists once, on a known rank).  "rank0": rank 0 combines everything, the others return None.
        wide: the context's bases are wide fixed-base tables (set_bases(..., precompute="wide")); the ranks share their VIRTUAL windows
        (msm_hip_launch_vwindows_batch_device: 8 at 19-bit digits), every window's record is a (weighted sum, plain total) pair and the finish is
        msm_hip_combine_vwindows_batch_curve."""
        assert 1 <= depth < self.SLOTS
        assert combine in ("all", "spread", "rank0")
        assert not (halves and wide)
        self.combine = combine
        self.depth = depth
        self.halves = halves
        self.wide = wide
        self.rec = 2 if wide else 1  # records per window in the gathered blocks
        if num_windows is None:
            num_windows = ctx.virtual_windows() if wide else NUM_WINDOWS // 2 if halves else NUM_WINDOWS
        assert num_windows > 0
        self.ctx, self.rank, self.world, self.group, self.num_windows = ctx, rank, world_size, group, num_windows
        self.w_begin, self.w_end = window_range(rank, world_size, num_windows)
        self.per = max_windows_per_rank(world_size, num_windows)
        # tuning aid: a single rank does the share rank 0 would have in a run of `emulate_world` ranks; its results are the
        # PARTIAL sums over those windows only (never a reported result)
        self.emulate = emulate_world if emulate_world > 1 and world_size == 1 else 0
        if self.emulate:
            self.w_begin, self.w_end = window_range(0, self.emulate, num_windows)
            self.per = max_windows_per_rank(self.emulate, num_windows)
        self.g = msms_per_issue
        assert self.g * self.per <= 64, "msms_per_issue x windows per rank must not exceed 64 local windows"
        dev = torch.device("cuda", ctx.device)
        rows = self.g * self.per * self.rec
        jb = ctx.jb  # bytes of a Jacobian record of the context's curve
        self.padded = [torch.zeros((rows, jb), dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.gathered = [torch.empty((world_size, rows, jb), dtype=torch.uint8, device=dev) for _ in range(self.SLOTS)]
        self.host = [torch.empty((world_size, rows, jb), dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self.host_np = [h.numpy() for h in self.host]  # views of the pinned buffers
        self.copied = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.nvec = [1] * self.SLOTS
        self.issued = 0
        self.completed = 0
        # MSM_SHARD_FORCE_COLLECTIVE=1: issue the RCCL all-gather even with a single rank (rehearsal of the multi-GPU
        # stream / queue layout on a one-GPU box; needs an initialised process group)
        self.collective = world_size > 1 or (os.environ.get("MSM_SHARD_FORCE_COLLECTIVE") == "1" and dist.is_initialized())

    def issue(self, scalars_dev, n=None, inputs_complete=False):
        """scalars_dev: one vector (CUDA uint8 [n, 32]) or, with `n` given, up to msms_per_issue contiguous vectors.
        inputs_complete: the scalars were complete (synchronised) before this call.  Otherwise the engine is ordered behind the
        current torch stream -- which also carries the previous launches' gather and copies (below), so that launch i+1 then
        starts only after launch i's gather: correct for scalars produced on that stream, but it serialises the pipeline."""
        assert self.issued - self.completed < self.depth, "pipeline full: call complete() first"
        slot = self.issued % self.SLOTS
        w_local = self.w_end - self.w_begin
        rows = scalars_dev.shape[0] if scalars_dev.dim() == 2 else scalars_dev.numel() // 32
        n = rows if n is None else n
        nvec = rows // n
        assert 1 <= nvec <= self.g and nvec * n == rows
        self.nvec[slot] = nvec
        if w_local > 0:
            launch = self.ctx.launch_vwindows_batch if self.wide else self.ctx.launch_half_windows_batch if self.halves else self.ctx.launch_windows_batch
            launch(scalars_dev, n, self.w_begin, self.w_end, slot, self.padded[slot][: nvec * w_local * self.rec], inputs_complete=inputs_complete)
            # gather + copies go to the CURRENT torch stream (normally the default stream): with GPU_MAX_HW_QUEUES=8 this
            # layout -- 3 engine streams, the default stream, RCCL's own -- keeps three launches in flight; a dedicated side
            # stream (or a 4th engine stream) was measured to collapse the pipeline to one at a time
            self.ctx.slot_wait_stream(slot)
        if self.collective:
            dist.all_gather_into_tensor(self.gathered[slot].view(-1), self.padded[slot].view(-1), group=self.group)
        else:
            self.gathered[slot].copy_(self.padded[slot].unsqueeze(0))
        self.host[slot].copy_(self.gathered[slot], non_blocking=True)
        self.copied[slot].record()
        self.issued += 1

    def complete(self):
        """Result of the oldest launch in flight: a G1, or the list of its G1 results when msms_per_issue > 1."""
        assert self.completed < self.issued
        slot = self.completed % self.SLOTS
        synced = False
        try:  # a launch that failed (device-side input error) is retired too: the pipeline stays usable
            self.copied[slot].synchronize()
            if self.w_end > self.w_begin:
                synced = True
                self.ctx.slot_sync(slot)
            out = self._combine(slot, self.nvec[slot])
        finally:
            self.completed += 1
            if not synced and self.w_end > self.w_begin:  # whatever failed above: the engine's slot is collected (never left busy)
                try:
                    self.ctx.slot_sync(slot)
                except Exception:
                    pass
        return out if self.g > 1 else out[0]

    def owner(self, v):
        """rank that combines vector v of a launch (None: every rank does)"""
        return None if self.combine == "all" or self.emulate else (0 if self.combine == "rank0" else v % self.world)

    def _combine(self, slot, nvec):
        """The launch's window sums (pinned host buffer: rank r's block holds [nvec][its windows] records) -> one G1 per MSM this rank
        owns (None for the others').  All Horner chains of the launch go through ONE library call (host pool: side by side)."""
        host = self.host_np[slot]
        if self.wide:  # a window's record is its (weighted sum, plain total) pair
            host = host.reshape(host.shape[0], host.shape[1] // 2, 2 * host.shape[2])
        if self.emulate:  # partial sums over this rank's windows only (tuning aid)
            nw = self.w_end - self.w_begin
            sums = np.ascontiguousarray(host[0, : nvec * nw])
        else:
            nw = self.num_windows
            sums = gathered_window_sums(host, nvec, self.world, nw)
        # (wide, emulated: the pairs of virtual windows 0 .. nw-1 only -- the partial result those windows give, as in the other modes)
        finish = MsmContext.combine_vwindows_batch if self.wide else MsmContext.combine_windows_batch
        mine = [v for v in range(nvec) if self.owner(v) in (None, self.rank)]
        if len(mine) == nvec:
            return finish(sums, nw, self.ctx.curve)
        out = [None] * nvec
        if mine:
            picked = np.ascontiguousarray(sums.reshape(nvec, nw, -1)[mine])
            for v, g in zip(mine, finish(picked, nw, self.ctx.curve)):
                out[v] = g
        return out
