"""Multi-GPU sharding of the hot path: one process per GPU, units = (UE, subframe).

The path partitions into independent units (a subframe of one UE never needs another's data; HARQ reuse is off at
rv 0), so the decode itself needs no collective: default placement is UE-major (UE u -> rank u mod world, keeping the
per-cell CRS tables and per-RNTI scrambling sequences local), and a batch of subframes of one UE can also be split
contiguously across ranks. The one exchange per batch is what the reference's worker pool does when its sf_workers hand
their decoded transport blocks to the single MAC (srsenb/src/phy/phy.cc:113-148, txrx.cc:105-134): `gather_results`
brings every rank's decoded TBs + CRC flags to rank 0 in ONE collective (RCCL gather over xGMI). The BLER
accounting afterwards is one all_reduce of a few counters per run.
"""


def result_layout(tb_stride, batch):
    """One rank's result record of a batch: [batch][tb_stride] transport-block bytes, then [batch] CRC flags, padded to 16 bytes.
    Returns (nbytes, offset of the flags)."""
    off = tb_stride * batch
    return (off + batch + 15) & ~15, off


def gather_results(result, gathered, dist=None, dst=0, async_op=False):
    """ONE collective per batch: every rank's `result` (1-D uint8 tensor, the record of result_layout) lands in row `rank` of
    `gathered` ([world, nbytes] on rank `dst`, None elsewhere). Device tensors with the nccl (= RCCL) backend, CPU tensors with gloo.
    Ordered after the work already queued on torch's current stream. Without a process group: a copy into row 0; a one-rank group takes
    the collective like any other (bench.py --force-dist: the RCCL calls on a one-GPU box)."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        if gathered is not None:
            gathered[0].copy_(result, non_blocking=True)
        return None
    rows = [gathered[r] for r in range(dist.get_world_size())] if dist.get_rank() == dst else None
    return dist.gather(result, rows, dst=dst, async_op=async_op)


def ue_for_rank(rank, base_rnti=0x1234, base_cell_id=1):
    """cfg4 of BASELINE.json: 8 UEs x 20 MHz, distinct RNTI 0x1234+u and cell id 1+u."""
    return {"rnti": base_rnti + rank, "cell_id": base_cell_id + rank}


def split_contiguous(n_units, world, rank):
    """[lo, hi) of `n_units` independent units owned by `rank`; sizes differ by at most one, nothing is dropped."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_counts(counts, dist=None, device=None):
    """Sum a list of integer counters over all ranks (no-op without an initialised process group)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(c) for c in counts]
    import torch
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=device)
    dist.all_reduce(t)
    return [int(v) for v in t.tolist()]
