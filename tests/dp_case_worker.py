"""Worker for tools/soak_dp.py: one rank of the library's data-parallel step (fmhip_dp_epoch) on GPU 0 over the host-staged
transport, for ONE randomly shaped case described by a JSON file (shards, model, batch size, cuts, exchange mode).
    python tests/dp_case_worker.py RANK CASE.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard(cfg, rank):
    from sparkfm_amd import synth
    rows = cfg["rows"][rank]
    if rows == 0:
        return dict(row_ptr=np.zeros(1, np.int64), col=np.zeros(0, np.int32), val=np.zeros(0, np.float32), y=np.zeros(0, np.float32))
    return synth.make_zipf(cfg["seed"], rows, cfg["n1_data"], cfg["lo"], cfg["hi"], zipf_s=1.05, row_begin=int(sum(cfg["rows"][:rank])))


def init(cfg):
    from sparkfm_amd import synth
    w0, w, v = synth.init_params(cfg["seed"] + 1, cfg["n1"], cfg["k"], stdev=0.05)
    w = np.random.default_rng(cfg["seed"] + 2).normal(0, 0.05, cfg["n1"])
    return 0.05, w, v


def main():
    rank, cfg = int(sys.argv[1]), json.load(open(sys.argv[2]))
    world = len(cfg["rows"])
    import torch.distributed as dist
    from sparkfm_amd import DataSet, FMModel
    from sparkfm_amd.distributed import HipDataParallelSGD, HostStagedComm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % cfg["port"], rank=rank, world_size=world)
    ds = DataSet.from_arrays(shard(cfg, rank), batch_rows=cfg["batch_rows"], device=0).cache()
    w0, w, v = init(cfg)
    fm = FMModel(cfg["n1"] - 1, cfg["k"], device=0)
    fm.w0, fm.w, fm.v = w0, w, v
    comm = HostStagedComm(fm, rank, world)
    dp = HipDataParallelSGD(comm, eta=cfg["eta"], regw=cfg["regw"], regv=cfg["regv"], exchange=cfg["exchange"],
                            upper_fractions=tuple(cfg["fractions"]))
    for _ in range(cfg["epochs"]):
        dp.learn(fm, ds)
    np.savez(cfg["out"] + ".%d.npz" % rank, w0=fm.w0, w=fm.w, v=fm.v, calls=np.array(getattr(comm, "calls", []), np.int64).reshape(-1, 2))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
