"""Multi-GPU sharding of the gmove path: one process per GPU, contiguous PAF-order shards, one exchange step.

Because the reference keeps the FIRST sample_limit accepted events of every k-mer in PAF-line order
(src/gmove.cpp:732, 925-927), rank g only needs to know how many accepted events ranks < g hold per k-mer:
    1. every rank:   counts_g = pg_count(shard_g)                       (uint64[n_slots], on the GPU)
    2. all ranks:    all_gather(counts)  -> base_g = sum_{h<g} counts_h  (RCCL over xGMI; 8 KB at k=5, 2 MB at k=9)
       meanwhile:    pg_stats()          median/MAD of every read of the shard (independent of the exchange)
    3. every rank:   pg_collect(base_g)  keeps the events whose global rank is < sample_limit
The per-k-mer stream of the whole job is the concatenation of the ranks' streams in rank order; freq.txt is
min(sum_g counts_g, sample_limit). Works with backend "nccl" (= RCCL, CUDA tensors) and "gloo" (CPU tensors).
"""
from typing import List, Tuple

import numpy as np


def shard_bounds(n_reads: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, PAF-ordered read range of `rank` (sizes differ by at most one read)."""
    q, r = divmod(n_reads, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def exchange_bases(counts, group=None):
    """counts: int64 tensor [n_slots] of this rank's accepted events (the bytes of pg_count's uint64 output).
    Returns (base, total): events held by lower ranks, and the job-wide total, per slot."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    flat = torch.empty(world * counts.numel(), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(flat, counts.contiguous().view(-1), group=group)
    allc = flat.view(world, -1)
    base = allc[:rank].sum(dim=0) if rank > 0 else torch.zeros_like(counts)
    return base, allc.sum(dim=0)


def merged_freq(total_counts, sample_limit: int, engine=None):
    """freq.txt of the whole job from the job-wide accepted counts. With the engine of an RCCL step (sharded_step on device
    shards) the column has already been produced on the device by pg_collect_gathered."""
    import torch
    if engine is not None and getattr(engine, "_job_totals", None) is not None and total_counts is engine._job_totals[0]:
        return engine._job_totals[1]
    return torch.clamp(total_counts, max=sample_limit)


def sharded_step(engine, shard, group=None, counts_buf=None, stream_ordered=False, gather_buf=None):
    """count -> exchange -> collect for one shard. Device-resident shards exchange on the GPU (RCCL); host
    shards exchange CPU tensors (gloo). Returns the job-wide accepted counts (on the RCCL path a tensor that aliases the
    engine's buffer: valid until the engine's next step or close()).
    stream_ordered: the engine runs on torch's current stream (GmoveEngine.use_torch_stream), so the
    collective is ordered by the stream and no host synchronisation is needed.
    gather_buf: optional preallocated int64 tensor [world * n_slots] on the shard's device (receive buffer of the
    all_gather). With RCCL the buffer goes straight into pg_collect_gathered, which sums the lower ranks' rows on the
    device together with the job-wide counts and the freq.txt column (engine.job_totals()): a step is count, ONE collective, collect."""
    import torch
    import torch.distributed as dist
    if shard.on_device:
        if counts_buf is None and gather_buf is not None and dist.get_backend(group) != "gloo":
            # this rank's row of the receive buffer: RCCL runs the all_gather in place (no local copy)
            r = dist.get_rank(group)
            counts_buf = gather_buf[r * engine.n_slots:(r + 1) * engine.n_slots]
        if counts_buf is None:
            counts_buf = torch.empty(engine.n_slots, dtype=torch.int64, device=shard.sig.device)
        engine.count(shard, out=counts_buf)
        gloo = dist.get_backend(group) == "gloo"
        if gloo or not stream_ordered:
            engine.sync()  # the library works on its own stream; the collective runs on torch's
        if gloo:  # rehearsal without RCCL: exchange through the host
            base, total = exchange_bases(counts_buf.cpu(), group)
            base = base.to(counts_buf.device)
            torch.cuda.current_stream().synchronize()
            engine.stats()
            engine.collect(base.contiguous())
        else:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            if gather_buf is None:
                gather_buf = torch.empty(world * counts_buf.numel(), dtype=counts_buf.dtype, device=counts_buf.device)
            # The statistics of every read (the streaming kernel, a third of the step) do not depend on the exchange: an
            # engine made with defer_stats queues them here, behind the ISSUE of the collective (which runs on RCCL's own
            # stream) and in front of the wait for it, so the all_gather's xGMI latency hides behind them.
            work = dist.all_gather_into_tensor(gather_buf, counts_buf.view(-1), group=group, async_op=True)
            engine.stats()
            work.wait()  # the current stream waits for the collective; the host does not
            if not stream_ordered:
                torch.cuda.current_stream().synchronize()
            engine.collect_gathered(gather_buf, world, rank)
            total = engine.job_totals()[0]  # summed by the kernel that computes this rank's base; [1] is the job's freq.txt column
    else:
        c = engine.count(shard)
        t = torch.from_numpy(c.view(np.int64).copy())
        base, total = exchange_bases(t, group)
        engine.stats()
        engine.collect(base.numpy().view(np.uint64).copy())
    return total


def gather_kept(counts, ev_len, samples, dst: int = 0, group=None):
    """The single-writer end of a multi-GPU job: every rank sends its kept events to rank `dst` (point-to-point over
    xGMI with RCCL, or gloo on CPU tensors), which returns the job's per-k-mer streams as the reference writes them --
    slot-major, and inside a slot the ranks' events in rank order (= PAF-line order).
    counts: int64[n_slots] kept events of this rank per slot; ev_len: int32[n_events] window lengths in this rank's
    slot-major order; samples: float64[n_samples] the windows back to back. Returns (counts, ev_len, samples) of the
    whole job on rank dst, None elsewhere. Output-side step: not part of the timed hot path."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = counts.device
    sizes = torch.tensor([ev_len.numel(), samples.numel()], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_counts = torch.empty(world * counts.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_counts, counts.contiguous().view(-1), group=group)
    all_sizes = all_sizes.view(world, 2).cpu(); all_counts = all_counts.view(world, -1)
    if rank != dst:
        if ev_len.numel(): dist.send(ev_len.contiguous(), dst, group=group)
        if samples.numel(): dist.send(samples.contiguous(), dst, group=group)
        return None
    lens, vals = [], []
    for r in range(world):
        ne, nsmp = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        if r == rank:
            lens.append(ev_len); vals.append(samples)
            continue
        le = torch.empty(ne, dtype=ev_len.dtype, device=dev); va = torch.empty(nsmp, dtype=samples.dtype, device=dev)
        if ne: dist.recv(le, r, group=group)
        if nsmp: dist.recv(va, r, group=group)
        lens.append(le); vals.append(va)
    n_slots = counts.numel()
    slots = torch.arange(n_slots, device=dev)
    ev_slot = torch.cat([torch.repeat_interleave(slots, all_counts[r]) for r in range(world)])  # rank-major
    order = torch.sort(ev_slot, stable=True).indices                                             # slot-major, rank order kept
    cat_len = torch.cat(lens).to(torch.int64)
    starts = torch.cumsum(cat_len, 0) - cat_len          # sample offset of every event in the rank-major concatenation
    out_len = cat_len[order]
    out_start = torch.cumsum(out_len, 0) - out_len
    cat_val = torch.cat(vals)
    total = int(out_len.sum())
    # sample i of the output belongs to output event e(i): source index = starts[order[e]] + (i - out_start[e])
    ev_of = torch.repeat_interleave(torch.arange(order.numel(), device=dev), out_len)
    src = starts[order][ev_of] + (torch.arange(total, device=dev) - out_start[ev_of])
    return all_counts.sum(dim=0), out_len.to(ev_len.dtype), cat_val[src]


def concat_rank_results(results: List["Result"]):
    """Per-slot concatenation of the ranks' kept events in rank order (what a single writer would dump)."""
    n_slots = results[0].counts.size
    counts = np.zeros(n_slots, np.uint64)
    vals = []
    for s in range(n_slots):
        vs = [r.slot_values(s) for r in results]
        vals.append(np.concatenate(vs) if vs else np.zeros(0))
        counts[s] = sum(int(r.counts[s]) for r in results)
    return counts, vals
