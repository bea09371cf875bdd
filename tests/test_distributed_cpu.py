"""World-size-2 data-parallel training on CPU ranks (gloo): the product's orchestration
(DataParallelSGD: shard -> local gradient -> ONE all-reduce(sum) -> identical update) with the
CPU oracle as the compute engine, against a single-process oracle run over the equivalent
global batches."""
import os
import socket
import subprocess
import sys

import numpy as np

import oracle
from helpers import random_problem

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    """A rendezvous for torch.distributed that cannot collide: a file store in a fresh directory (a port found by binding to 0
    and closing can be taken by someone else before the ranks bind it again — seen once on a GPU box: EADDRINUSE)."""
    import tempfile
    return "file://" + os.path.join(tempfile.mkdtemp(prefix="fmhip_rdzv_"), "store")


def test_two_ranks_equal_one_process(tmp_path):
    port, out = str(free_port()), str(tmp_path / "dp")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), "2", port, out], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    # replicas are bit-identical after training
    np.testing.assert_array_equal(r0["v"], r1["v"])
    np.testing.assert_array_equal(r0["w"], r1["w"])
    assert float(r0["w0"]) == float(r1["w0"])
    # rank 0 has 80 rows (3 batches of 32), rank 1 has 150 (5 batches): 5 global steps per epoch
    assert list(r0["steps"]) == [5, 5] and list(r1["steps"]) == [5, 5]
    # single-process equivalent: global batch j = rank0's batch j  U  rank1's batch j
    a = random_problem(2024, 230, 60, 5, 0, 12, empty_rows=(4,))
    shards = [(0, 80), (80, 230)]
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(2):
        for j in range(5):
            rows = []
            for lo, hi in shards:
                b0, b1 = lo + j * 32, min(hi, lo + (j + 1) * 32)
                rows.extend(range(b0, max(b0, b1)))
            rp = np.zeros(len(rows) + 1, np.int64)
            cols, vals = [], []
            for i, r in enumerate(rows):
                s = slice(a["row_ptr"][r], a["row_ptr"][r + 1])
                cols.append(a["col"][s])
                vals.append(a["val"][s])
                rp[i + 1] = rp[i] + (s.stop - s.start)
            w0, w, v, _ = oracle.sgd_step(w0, w, v, 0, len(rows), rp, np.concatenate(cols), np.concatenate(vals),
                                          a["y"][rows], 0.05, 0.01, 0.02, 0.03)
    np.testing.assert_allclose(r0["v"], v, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(r0["w"], w, rtol=1e-11, atol=1e-13)
    assert abs(float(r0["w0"]) - w0) < 1e-12
