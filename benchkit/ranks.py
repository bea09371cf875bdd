"""Multi-rank launch and control plane of bench.py: rendezvous, barriers and timer reductions — never the gradients."""
import subprocess
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_PG_GENERATION = [0]


def init_process_group(dist, backend, **kw):
    """torch.distributed's rendezvous: the launcher's env:// (torch.distributed.run sets MASTER_*), or — ranks spawned by this file —
    a file store, one file per process group this run creates (the fallback exchange makes a second one)."""
    rdzv = os.environ.get("FMHIP_BENCH_RDZV")
    if rdzv:
        _PG_GENERATION[0] += 1
        return dist.init_process_group(backend, init_method="%s.%d" % (rdzv, _PG_GENERATION[0]), rank=int(os.environ["RANK"]),
                                       world_size=int(os.environ["WORLD_SIZE"]), **kw)
    return dist.init_process_group(backend, **kw)


class TorchCtl:
    """The bench's control plane over torch.distributed (gloo; nccl when the exchange itself is torch's): barriers and
    reductions of a few timers — never the gradients."""

    def __init__(self, dist, torch, on_gpu):
        self.dist, self.torch, self.on_gpu = dist, torch, on_gpu

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, values, op="max"):
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64)
        if self.on_gpu:
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(x) for x in t.cpu()]

    def sum_counts(self, counts):
        t = self.torch.from_numpy(counts)
        if self.on_gpu:
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()


class ThreadCtl:
    """The same over the ranks-as-threads group (--transport threads)."""

    def __init__(self, group, rank):
        self.group, self.rank = group, rank

    def barrier(self):
        self.group.barrier()

    def allreduce(self, values, op="max"):
        return self.group.allreduce(self.rank, values, op)

    def sum_counts(self, counts):
        return sum(self.group.exchange(self.rank, counts))


class NoCtl:
    def barrier(self):
        pass

    def allreduce(self, values, op="max"):
        return [float(v) for v in values]

    def sum_counts(self, counts):
        return counts


def spawn_ranks(args, argv, log_dir=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this process
    has not touched the GPU and never will), forward rank 0's JSON lines AS THEY COME (a killed run keeps what was
    written), exit with the worst exit code.  Every rank's stderr goes to <log_dir>/rank<r>.log (rank 0's to this
    process's stderr as well); the tail of a failed rank's log is shown."""
    import tempfile
    import threading
    log_dir = log_dir or tempfile.mkdtemp(prefix="fmhip_bench_logs_")
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    # the ranks meet through a file store in a fresh directory: a port found by binding to 0 and closing it can be taken by
    # someone else before rank 0 binds it again (EADDRINUSE, seen once on a GPU box)
    rdzv = "file://" + os.path.join(tempfile.mkdtemp(prefix="fmhip_bench_rdzv_"), "store")
    bench_py = os.path.join(ROOT, "bench.py")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), FMHIP_BENCH_RDZV=rdzv, FMHIP_BENCH_LOG_DIR=log_dir,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(open(os.path.join(log_dir, "rank%d.log" % r), "wb"))
        procs.append(subprocess.Popen([sys.executable, bench_py] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=logs[-1]))

    def forward():       # rank 0's lines go out as soon as they are complete
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()
    fw = threading.Thread(target=forward, daemon=True)
    fw.start()
    # a rank that dies (no such GPU, out of memory ...) must not leave the others waiting in a collective
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:
            rc = abs(failed[0].returncode) or 1
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    fw.join(timeout=10)
    for r, p in enumerate(procs):
        rc = max(rc, abs(p.returncode or 0))
        logs[r].close()
        if p.returncode or r == 0:
            try:
                tail = open(os.path.join(log_dir, "rank%d.log" % r), "rb").read()[-(4000 if p.returncode else 1500):].decode(errors="replace")
            except OSError:
                tail = ""
            if tail:
                sys.stderr.write("[bench] rank %d (rc %s), end of %s/rank%d.log:\n%s\n" % (r, p.returncode, log_dir, r, tail))
    raise SystemExit(rc)
