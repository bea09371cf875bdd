#!/usr/bin/env python3
"""Soak of the data-parallel step: randomly shaped cases — 2 or 3 real ranks on ONE GPU over the host-staged transport
(fmhip_comm_create_external; RCCL refuses two ranks on one device), uneven and empty shards, model widths from narrower than a
batch to far wider, 0-4 cuts, the three exchange modes, weight decay on and off — each compared with the fp64 oracle over
the equivalent global batches; replicas must agree bit for bit and issue the same collectives.
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
import oracle  # noqa: E402
import dp_case_worker as W  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-9))


def make_case(seed):
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 2, 3]))
    n1_data = int(rng.choice([40, 300, 800, 3000]))
    n1 = int(n1_data * rng.choice([1, 1, 4, 50]))
    rows = [int(rng.choice([0, 150, 700, 1700, 3000])) for _ in range(world)]
    if sum(rows) == 0:
        rows[int(rng.integers(0, world))] = 900
    hi = int(rng.integers(2, min(n1_data, 40) + 1))
    cfg = dict(seed=int(1000 + seed), rows=rows, n1_data=n1_data, n1=n1, k=int(rng.choice([4, 16, 32, 64])), lo=int(rng.integers(1, hi + 1)), hi=hi,
               batch_rows=int(rng.choice([64, 300, 1000])), exchange=str(rng.choice(["dense", "sharded", "touched"])),
               fractions=[[], [0.3], [0.1, 0.4], [0.05, 0.15, 0.3, 0.55]][int(rng.integers(0, 4))], epochs=int(rng.choice([1, 2])),
               eta=0.02, regw=float(rng.choice([0.0, 1e-3])), regv=float(rng.choice([0.0, 1e-3])))
    return cfg


def oracle_epochs(cfg):
    shards = [W.shard(cfg, r) for r in range(len(cfg["rows"]))]
    w0, w, v = W.init(cfg)
    br = cfg["batch_rows"]
    steps = max((len(d["y"]) + br - 1) // br for d in shards)
    for _ in range(cfg["epochs"]):
        for j in range(steps):                                    # lock-step: every rank's j-th batch (or nothing) forms the global batch
            rp, cols, vals, ys = [0], [], [], []
            for d in shards:
                n = len(d["y"])
                for r in range(j * br, min(n, (j + 1) * br)):
                    a, b = d["row_ptr"][r], d["row_ptr"][r + 1]
                    cols.append(d["col"][a:b])
                    vals.append(d["val"][a:b].astype(np.float64))
                    rp.append(rp[-1] + (b - a))
                    ys.append(float(d["y"][r]))
            if not ys:
                continue
            w0, w, v, _ = oracle.sgd_step(w0, w, v, 0, len(ys), np.array(rp, np.int64), np.concatenate(cols) if cols else np.zeros(0, np.int32),
                                          np.concatenate(vals) if vals else np.zeros(0), np.array(ys), cfg["eta"], 0.0, cfg["regw"], cfg["regv"])
    return w0, w, v


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
        procs = [subprocess.Popen([sys.executable, worker, str(r), path]) for r in range(world)]
        rcs = [p.wait(timeout=600) for p in procs]
        tag = "case %d %s" % (seed, {k_: cfg[k_] for k_ in ("rows", "n1_data", "n1", "k", "lo", "hi", "batch_rows", "exchange", "fractions", "epochs", "regw", "regv")})
        assert rcs == [0] * world, (tag, rcs)
        r0 = np.load(cfg["out"] + ".0.npz")
        for r in range(1, world):
            r1 = np.load(cfg["out"] + ".%d.npz" % r)
            assert np.array_equal(r0["v"], r1["v"]) and np.array_equal(r0["w"], r1["w"]) and float(r0["w0"]) == float(r1["w0"]), (tag, "replicas differ")
            assert np.array_equal(r0["calls"], r1["calls"]), (tag, "the ranks issued different collectives")
        w0, w, v = oracle_epochs(cfg)
        ev, ew = rel(r0["v"], v), rel(r0["w"], w)
        assert ev <= 1e-5 and ew <= 1e-5 and abs(float(r0["w0"]) - w0) <= 1e-5 * abs(w0) + 1e-6, (tag, ev, ew, float(r0["w0"]), w0)
        print("ok %s  (rel err V %.1e w %.1e, %d collectives)" % (tag, ev, ew, len(r0["calls"])), flush=True)


if __name__ == "__main__":
    main()
