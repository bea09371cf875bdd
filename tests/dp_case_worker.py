"""Worker for the data-parallel GPU tests and tools/soak_dp.py: ONE case (tests/dp_cases.py: shards, model, batch size, cuts,
exchange mode) through the library's own step on GPU 0.
    python tests/dp_case_worker.py RANK CASE.json       one rank per PROCESS, host-staged gloo transport (<= 6 processes per card)
    python tests/dp_case_worker.py threads CASE.json    every rank a THREAD of this process (a world of 8 on one GPU); checks the
                                                        case against the oracle itself and prints one JSON summary line"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dp_cases  # noqa: E402
from dp_cases import init, shard  # noqa: E402,F401  (tools/soak_dp.py reads them from here)


def main():
    cfg = json.load(open(sys.argv[2]))
    world = len(cfg["rows"])
    if sys.argv[1] == "threads":
        res = dp_cases.run_threads(cfg)
        ev, ew = dp_cases.check(cfg, res)
        r0 = res[0]
        print(json.dumps(dict(world=world, rel_err_v=ev, rel_err_w=ew, calls=r0["calls"].tolist(), cuts=r0["cuts"].tolist(), rows=int(r0["rows"]),
                              steps=int(r0["steps"]), info=r0["info"])))
        return
    rank = int(sys.argv[1])
    import torch.distributed as dist
    from sparkfm_amd.distributed import HostStagedComm
    dist.init_process_group("gloo", init_method=(cfg["port"] if "://" in str(cfg["port"]) else "tcp://127.0.0.1:%s" % cfg["port"]), rank=rank, world_size=world)
    r = dp_cases.run_rank(cfg, rank, lambda fm: HostStagedComm(fm, rank, world), dist.barrier)
    np.savez(cfg["out"] + ".%d.npz" % rank, w0=r["w0"], w=r["w"], v=r["v"], calls=r["calls"], cuts=r["cuts"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
