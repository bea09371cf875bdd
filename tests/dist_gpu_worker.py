"""Worker for test_two_ranks_on_one_gpu: one rank of the REAL data-parallel path (HipEngine, the
feature-chunked backward, async all-reduces) — both ranks share cuda:0, the collective runs over gloo
(which moves CUDA tensors through the host), so the orchestration is exercised end to end on a
one-GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out, overlap = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1"
    k = int(sys.argv[6]) if len(sys.argv) > 6 else 32
    import torch
    import torch.distributed as dist
    from sparkfm_amd import DataSet, FMModel, synth
    from sparkfm_amd.distributed import DataParallelSGD, torch_stream_handle
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=(port if "://" in str(port) else "tcp://127.0.0.1:%s" % port), rank=rank, world_size=world)
    # the same virtual dataset as the single-process reference: rank r owns rows [r*3000, (r+1)*3000)
    # uneven shards: rank 1 has 2 batches (1000 + 700 rows) against rank 0's 3, so its last step contributes zeros
    d = synth.make_zipf(77, 3000 if rank == 0 else 1700, 800, 4, 24, zipf_s=1.05, row_begin=rank * 3000)
    ds = DataSet.from_arrays(d, batch_rows=1000, device=0).cache()
    w0, w, v = synth.init_params(5, 800, k, stdev=0.05)
    w = np.random.default_rng(9).normal(0, 0.05, 800)
    fm = FMModel(799, k, device=0, stream=torch_stream_handle(0))
    fm.w0, fm.w, fm.v = w0, w, v
    dp = DataParallelSGD(eta=0.05, regw=1e-3, regv=1e-3, overlap=overlap)
    for _ in range(2):
        dp.learn(fm, ds)
    torch.cuda.synchronize()
    np.savez(out + ".%d.npz" % rank, w0=fm.w0, w=fm.w, v=fm.v, cuts=np.array(dp.cuts if dp.cuts else [-1]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
