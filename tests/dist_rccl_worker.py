"""Worker for test_library_side_exchange_two_ranks*: the data-parallel step runs INSIDE the library (fmhip_dp_epoch).
Transport "rccl": one rank = one GPU, fmhip_comm_create over RCCL; torch.distributed (gloo) only ships the 128-byte
unique id and the final barrier.  Transport "host": both ranks on GPU 0, fmhip_comm_create_external with a
host-staged gloo all-reduce (RCCL refuses two ranks on one device) — same schedule, cuts, row counts and update."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    transport = sys.argv[5] if len(sys.argv) > 5 else "rccl"
    fractions = tuple(float(x) for x in sys.argv[6].split(",") if x) if len(sys.argv) > 6 else None
    exchange = sys.argv[7] if len(sys.argv) > 7 else "dense"
    n1 = int(sys.argv[8]) if len(sys.argv) > 8 else 800         # a model wider than the data exercises the rows-only update
    dev = 0 if transport == "host" else rank
    import torch.distributed as dist
    from sparkfm_amd import DataSet, FMModel, synth
    from sparkfm_amd.distributed import HipDataParallelSGD, HostStagedComm, RcclComm
    dist.init_process_group("gloo", init_method=(port if "://" in str(port) else "tcp://127.0.0.1:%s" % port), rank=rank, world_size=world)
    # uneven shards: rank 1 has 2 batches against rank 0's 3; a third rank has no rows at all
    if rank < 2:
        d = synth.make_zipf(77, 3000 if rank == 0 else 1700, 800, 4, 24, zipf_s=1.05, row_begin=rank * 3000)
    else:
        d = dict(row_ptr=np.zeros(1, np.int64), col=np.zeros(0, np.int32), val=np.zeros(0, np.float32), y=np.zeros(0, np.float32))
    ds = DataSet.from_arrays(d, batch_rows=1000, device=dev).cache()
    w0, w, v = synth.init_params(5, n1, 32, stdev=0.05)
    w = np.random.default_rng(9).normal(0, 0.05, n1)
    fm = FMModel(n1 - 1, 32, device=dev)
    fm.w0, fm.w, fm.v = w0, w, v
    comm = HostStagedComm(fm, rank, world) if transport == "host" else RcclComm(fm, rank, world)
    kw = {} if fractions is None else {"upper_fractions": fractions}
    dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, exchange=exchange, **kw)
    for _ in range(2):
        dp.learn(fm, ds)
    calls = np.array(getattr(comm, "calls", []), np.int64).reshape(-1, 2)
    np.savez(out + ".%d.npz" % rank, w0=fm.w0, w=fm.w, v=fm.v, cuts=np.array(dp.cuts), calls=calls,
             rows=dp.last_stats["rows"], steps=dp.last_stats["steps"], mean_union=dp.exchange_info()["mean_union_rows"])
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
