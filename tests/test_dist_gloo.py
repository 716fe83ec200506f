"""world_size-2 rehearsal of the N>1 path on CPU (gloo): UE-per-rank placement, contiguous splitting, the per-batch gather of decoded
transport blocks + CRC flags to rank 0 (srslte-emane_amd/sharding.py:gather_results, the one collective of a batch) and the accounting
all_reduce. The device call is stood in for by the oracle chain - this test is about the sharding, not the kernels."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("srslte-emane_amd.sharding")
    from lte_sim import DlConfig, make_subframe, oracle_rx
    ue = sh.ue_for_rank(rank)
    cfg = DlConfig(6, ue["cell_id"], 1, 936, rnti=ue["rnti"])
    rng = np.random.default_rng(100 + rank)
    good = bad = 0
    # this rank's result record of the batch, laid out as bench.py lays it out: [batch][tb_stride] TB bytes, then [batch] CRC flags
    batch, tb_stride = 3, (936 // 8 + 6 + 15) & ~15
    nbytes, ok_off = sh.result_layout(tb_stride, batch)
    rec = np.zeros(nbytes, np.uint8)
    sent = []
    for b, t in enumerate((1, 2, 3)):
        iq, data = make_subframe(cfg, t, rng, snr_db=12.0 if rank == 0 else -6.0)
        r = oracle_rx(cfg, iq, t)
        good += int(r["ok"] and np.array_equal(r["tb"][:117], data))
        bad += int(not r["ok"])
        rec[b * tb_stride:b * tb_stride + 120] = r["tb"]
        rec[ok_off + b] = int(r["ok"])
        sent.append(data)
    # ONE collective for the batch: every rank's record to rank 0, rank order = row order
    gathered = torch.zeros((world, nbytes), dtype=torch.uint8) if rank == 0 else None
    sh.gather_results(torch.from_numpy(rec), gathered, dist)
    recs = [None] * world
    dist.all_gather_object(recs, (rec.tobytes(), [d.tobytes() for d in sent]))  # what each rank held, for rank 0's placement check only
    tot = sh.reduce_counts([good, bad, 3, ue["rnti"]], dist)
    lo, hi = sh.split_contiguous(13, world, rank)
    spans = [None] * world
    dist.all_gather_object(spans, (lo, hi))
    if rank == 0:
        placed = all(gathered[r].numpy().tobytes() == recs[r][0] for r in range(world))
        distinct = recs[0][0] != recs[1][0]
        # rank 0 (12 dB) decoded its UE's three blocks: they sit in row 0 at their subframe's offset with the flag set
        row0 = gathered[0].numpy()
        tb_ok = all(row0[b * tb_stride:b * tb_stride + 117].tobytes() == recs[0][1][b] and row0[ok_off + b] == 1 for b in range(batch))
        row1_flags = gathered[1].numpy()[ok_off:ok_off + batch].tolist()
        out.put((tot, spans, good, bad, placed, distinct, tb_ok, row1_flags))
    dist.destroy_process_group()


def test_two_rank_sharding_and_accounting():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p_ in procs:
        p_.start()
    tot, spans, good0, bad0, placed, distinct, tb_ok, row1_flags = out.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert tot[2] == 6 and tot[3] == 0x1234 + 0x1235          # every rank contributed, distinct UEs
    assert tot[0] + tot[1] == 6 and good0 == 3 and tot[1] == 3  # rank 0 (12 dB) decodes all, rank 1 (-6 dB) none
    assert spans == [(0, 7), (7, 13)]                          # contiguous, complete, balanced
    assert placed and distinct and tb_ok and row1_flags == [0, 0, 0]  # gather: rank r's record in row r, TBs and flags where the layout says


def test_gather_results_without_a_process_group():
    sh = importlib.import_module("srslte-emane_amd.sharding")
    nbytes, off = sh.result_layout(9440, 128)
    assert off == 9440 * 128 and nbytes % 16 == 0 and nbytes >= off + 128
    rec, g = torch.arange(64, dtype=torch.uint8), torch.zeros((1, 64), dtype=torch.uint8)
    assert sh.gather_results(rec, g, None) is None and torch.equal(g[0], rec)


def test_split_contiguous_properties():
    sh = importlib.import_module("srslte-emane_amd.sharding")
    for n in (0, 1, 7, 128, 129, 1664):
        for world in (1, 2, 3, 4, 8):
            spans = [sh.split_contiguous(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
