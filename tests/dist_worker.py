"""Worker for tests/test_distributed_cpu.py: one CPU rank of the data-parallel trainer.

The compute engine here is the CPU ORACLE (allowed: this file lives in tests/), plugged into the
product's orchestration (sparkfm_amd.distributed.DataParallelSGD) so that the sharding, the step
schedule, the zero-contribution of exhausted ranks and the all-reduce are exercised with gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """Packed gradient = [gv (k*n1) | gw (n1) | g0, sse, rows] as a CPU torch tensor."""

    def __init__(self, fm, shard):
        import torch
        self.fm, self.s = fm, shard
        self.k, self.n1 = fm["v"].shape
        self.grad = torch.zeros(self.k * self.n1 + self.n1 + 3, dtype=torch.float64)
        self.batch_rows = shard["batch_rows"]
        n_rows = len(shard["y"])
        self.n_batches = -(-n_rows // self.batch_rows) if n_rows else 0

    def compute(self, b):
        import oracle
        import torch
        s, fm = self.s, self.fm
        r0, r1 = b * self.batch_rows, min(len(s["y"]), (b + 1) * self.batch_rows)
        gv, gw, g0, sse, _ = oracle.batch_grad(fm["w0"], fm["w"], fm["v"], r0, r1, s["row_ptr"], s["col"], s["val"], s["y"])
        self.grad[:] = torch.from_numpy(np.concatenate([gv.T.reshape(-1), gw, [g0, sse, r1 - r0]]))

    def compute_empty(self):
        self.grad.zero_()

    def apply(self, eta, reg0, regw, regv):
        g = self.grad.numpy()
        k, n1, fm = self.k, self.n1, self.fm
        rows = g[-1]
        gv = g[:k * n1].reshape(n1, k).T
        gw = g[k * n1:k * n1 + n1]
        fm["w0"] = fm["w0"] - eta * (g[-3] / rows + reg0 * fm["w0"])
        fm["w"] = fm["w"] - eta * (gw / rows + regw * fm["w"])
        fm["v"] = fm["v"] - eta * (gv / rows + regv * fm["v"])
        self.last = dict(sse=float(g[-2]), rows=int(rows))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch.distributed as dist
    from sparkfm_amd.distributed import DataParallelSGD, shard_rows
    from helpers import random_problem
    dist.init_process_group("gloo", init_method=(port if "://" in str(port) else "tcp://127.0.0.1:%s" % port), rank=rank, world_size=world)
    a = random_problem(2024, 230, 60, 5, 0, 12, empty_rows=(4,))
    lo, hi = shard_rows(230, rank, world)
    if rank == world - 1:
        hi = 230
    # rank 0 gets fewer rows than rank 1 on purpose when world == 2: uneven step counts
    if world == 2:
        lo, hi = (0, 80) if rank == 0 else (80, 230)
    p0, p1 = a["row_ptr"][lo], a["row_ptr"][hi]
    shard = dict(row_ptr=a["row_ptr"][lo:hi + 1] - p0, col=a["col"][p0:p1], val=a["val"][p0:p1], y=a["y"][lo:hi],
                 batch_rows=32)
    fm = dict(w0=a["w0"], w=a["w"].copy(), v=a["v"].copy())
    dp = DataParallelSGD(eta=0.05, reg0=0.01, regw=0.02, regv=0.03, engine_factory=OracleEngine)
    steps = []
    for _ in range(2):
        eng = dp.engine(fm, shard)
        steps.append(dp.global_steps(eng))
        dp.learn(fm, shard)
    np.savez(out + ".%d.npz" % rank, w0=fm["w0"], w=fm["w"], v=fm["v"], steps=steps)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
