#!/usr/bin/env python3
"""Soak of the data-parallel step: randomly shaped cases — 2 to 8 real ranks on ONE GPU (2-3 as processes over the host-staged
gloo transport, 4 and 8 as threads of one process over ThreadStagedComm: RCCL refuses two ranks on one device and the test
pool admits at most 6 processes on the card), uneven and empty shards, model widths from narrower than a batch to far
wider, 0-4 cuts, the four exchange modes, weight decay on and off, ascending and permuted batch order — each compared
with the fp64 oracle over the equivalent global batches; replicas must agree bit for bit and issue the same collectives.
    python3 tools/soak_dp.py [cases, default 12] [first seed, default 1]"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_cases  # noqa: E402


def free_port():
    """A rendezvous for torch.distributed that cannot collide: a file store in a fresh directory (a port found by binding to 0
    and closing can be taken by someone else before the ranks bind it again — seen once on a GPU box: EADDRINUSE)."""
    import tempfile
    return "file://" + os.path.join(tempfile.mkdtemp(prefix="fmhip_rdzv_"), "store")


def make_case(seed):
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 2, 3, 4, 8, 8]))
    n1_data = int(rng.choice([40, 300, 800, 3000]))
    n1 = int(n1_data * rng.choice([1, 1, 4, 50])) + int(rng.choice([0, 0, 1, 3, 5]))      # n+1 is often no multiple of the world
    rows = [int(rng.choice([0, 150, 700, 1700, 3000])) for _ in range(world)]
    if sum(rows) == 0:
        rows[int(rng.integers(0, world))] = 900
    hi = int(rng.integers(2, min(n1_data, 40) + 1))
    cfg = dict(seed=int(1000 + seed), rows=rows, n1_data=n1_data, n1=n1, k=int(rng.choice([4, 16, 32, 64])), lo=int(rng.integers(1, hi + 1)), hi=hi,
               batch_rows=int(rng.choice([64, 300, 1000])), exchange=str(rng.choice(["dense", "sharded", "touched", "pipelined"])),
               fractions=[[], [0.3], [0.1, 0.4], [0.05, 0.15, 0.3, 0.55]][int(rng.integers(0, 4))], epochs=int(rng.choice([1, 2])),
               eta=0.02, regw=float(rng.choice([0.0, 1e-3])), regv=float(rng.choice([0.0, 1e-3])),
               shuffle_seed=None if rng.random() < 0.6 else int(rng.integers(0, 1000)))
    if rng.random() < 0.35:
        cfg["reverse_ids"] = True          # the frequent features carry the HIGHEST ids (round 5: the hot block above the plan's cuts)
    return cfg


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    worker = os.path.join(ROOT, "tests", "dp_case_worker.py")
    for seed in range(first, first + n_cases):
        cfg = make_case(seed)
        tmp = tempfile.mkdtemp()
        cfg["port"], cfg["out"] = free_port(), os.path.join(tmp, "dp")
        path = os.path.join(tmp, "case.json")
        json.dump(cfg, open(path, "w"))
        world = len(cfg["rows"])
        tag = "case %d %s" % (seed, {k_: cfg[k_] for k_ in ("rows", "n1_data", "n1", "k", "lo", "hi", "batch_rows", "exchange", "fractions", "epochs", "regw", "regv",
                                                            "shuffle_seed")}) + (" reverse_ids" if cfg.get("reverse_ids") else "")
        if world > 3:
            r = subprocess.run([sys.executable, worker, "threads", path], stdout=subprocess.PIPE, timeout=900)
            assert r.returncode == 0, tag
            s = json.loads(r.stdout.decode().strip().splitlines()[-1])
            print("ok %s  (threads; rel err V %.1e w %.1e, %d collectives)" % (tag, s["rel_err_v"], s["rel_err_w"], len(s["calls"])), flush=True)
            continue
        procs = [subprocess.Popen([sys.executable, worker, str(r), path]) for r in range(world)]
        rcs = [p.wait(timeout=600) for p in procs]
        assert rcs == [0] * world, (tag, rcs)
        res = [dict(np.load(cfg["out"] + ".%d.npz" % r)) for r in range(world)]
        ev, ew = dp_cases.check(cfg, res)
        print("ok %s  (rel err V %.1e w %.1e, %d collectives)" % (tag, ev, ew, len(res[0]["calls"])), flush=True)


if __name__ == "__main__":
    main()
