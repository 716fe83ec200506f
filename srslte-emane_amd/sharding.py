"""Multi-GPU sharding of the hot path: one process per GPU, units = (UE, subframe).

The path partitions into independent units (a subframe of one UE never needs another's data; HARQ reuse is off at
rv 0), so there is NO data-path collective: default placement is UE-major (UE u -> rank u mod world, keeping the
per-cell CRS tables and per-RNTI scrambling sequences local), and a batch of subframes of one UE can also be split
contiguously across ranks. The only exchange is the BLER accounting: one all_reduce of four counters per run.
"""


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
