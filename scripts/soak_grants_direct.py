#!/usr/bin/env python3
"""Soak of the decoders' own transport-block assembly for ragged batches (tdec_set_tb_ragged: per-slot atomics, the last block to arrive gives the
verdict; a block kept from an earlier transmission contributes its stored bytes) under concurrency: eight pipeline objects on eight streams decode
four different mixed-grant batches, each followed by the retransmission of the same data, over and over without a host synchronisation in between;
after every round each object's transport blocks and verdicts must equal what ONE object produced serially.
  GPU_MAX_HW_QUEUES=8 python scripts/soak_grants_direct.py [rounds]"""
import importlib
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    hp = importlib.import_module("srslte-emane_amd")
    from lte_sim import DlConfig, make_subframe
    L = hp.lib()
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    P, cell_id, nsf, nobj, nbatch = 50, 9, 24, 8, 4
    # (first PRB, PRBs, mod, tbs, snr): every decoder kind, one- to three-block transport blocks, SNRs around the thresholds (failures included)
    kinds = [(0, 3, 1, 296, 0.5), (5, 5, 1, 616, 1.0), (0, 12, 1, 1544, 1.0), (0, 25, 2, 4008, 6.3), (0, 50, 2, 12216, 8.2), (0, 50, 2, 15264, 9.5), (3, 30, 2, 6200, 8.5)]
    batches = []
    for bi in range(nbatch):
        rng = np.random.default_rng(900 + bi)
        iqs, grants, nbytes, iqs2, grants2 = [], [], [], [], []
        for b in range(nsf):
            first, n, mod, tbs, snr = kinds[int(rng.integers(0, len(kinds)))]
            mask = np.zeros((2, P), np.uint8)
            mask[:, first:first + n] = 1
            cfg = DlConfig(P, cell_id, mod, tbs, cfi=1 + b % 3, rnti=0x400 + b, prb_mask=mask)
            iq, data = make_subframe(cfg, b, rng, snr_db=snr + float(rng.uniform(-0.7, 0.7)))
            iqs.append(iq)
            grants.append(hp.DlGrant.make(P, mod, tbs, cfg.rnti, cfi=cfg.cfi, prb_mask=mask))
            nbytes.append(tbs // 8 + 3)
            # the retransmission of the same data (rv 2), ten subframes later: blocks that passed are kept, delivered transport blocks refused
            iqs2.append(make_subframe(cfg, 10 + b, rng, snr_db=snr + 0.5, rv=2, data=data)[0])
            grants2.append(hp.DlGrant.make(P, mod, tbs, cfg.rnti, cfi=cfg.cfi, rv=2, new_data=False, prb_mask=mask))
        batches.append((hp.DevBuf.from_host(np.ascontiguousarray(np.stack(iqs), np.complex64)), (hp.DlGrant * nsf)(*grants), nbytes,
                        hp.DevBuf.from_host(np.ascontiguousarray(np.stack(iqs2), np.complex64)), (hp.DlGrant * nsf)(*grants2)))
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rxs = [hp.DlRx(cell_id, P, 1, 0, 1, 15264, 6, nsf, True, hc) for _ in range(nobj)]
    streams = [L.srslte_hip_stream_create() for _ in range(nobj)]
    ref = []
    def rows(rx, nbytes):  # the transport blocks proper (a row keeps bytes of longer blocks of earlier calls behind them)
        tb = rx.d_tb.to_host(np.uint8).reshape(-1, rx.tb_stride)
        return [tb[b, :nbytes[b]].copy() for b in range(nsf)]

    def call(s, din, arr, tti0):
        assert L.srslte_hip_dl_rx_batch_grants(rxs[s].h, din.ptr, tti0, nsf, arr, rxs[s].d_tb.ptr, rxs[s].tb_stride, rxs[s].d_ok.ptr, streams[s]) == 0

    first = []
    for din, arr, nbytes, din2, arr2 in batches:  # serial reference: one object, one stream, a synchronisation behind every call
        call(0, din, arr, 0)
        hp.sync()
        first.append(int(rxs[0].d_ok.to_host(np.uint8)[:nsf].sum()))
        call(0, din2, arr2, 10)
        hp.sync()
        ref.append((rows(rxs[0], nbytes), rxs[0].d_ok.to_host(np.uint8)[:nsf].copy()))
    print("reference: delivered by the first transmission", first, "by the retransmission", [int(r[1].sum()) for r in ref], "of", nsf)
    bad = 0
    for it in range(rounds):
        for rep in range(2):  # two transmissions and their retransmissions per object queue up before the host looks
            for s in range(nobj):
                din, arr, _, din2, arr2 = batches[(it + s + rep) % nbatch]
                call(s, din, arr, 0)
                call(s, din2, arr2, 10)
        hp.sync()
        for s in range(nobj):
            tb, ok = ref[(it + s + 1) % nbatch]
            got = rows(rxs[s], batches[(it + s + 1) % nbatch][2])
            if not (np.array_equal(rxs[s].d_ok.to_host(np.uint8)[:nsf], ok) and all(np.array_equal(g, t) for g, t in zip(got, tb))):
                bad += 1
                print("MISMATCH round %d object %d" % (it, s))
    print("soak: %d rounds x %d objects x 4 calls, %d mismatches" % (rounds, nobj, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
