"""rocprofv3 --pmc counter passes run by bench.py as child processes, and the committed fallback (profiles/pmc_traffic.json)."""
import json
import subprocess
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def committed_pmc(config, k, batch_rows):
    """Counter-derived figures of the committed rocprofv3 --pmc passes for this configuration
    (profiles/pmc_traffic.json): {kernel: {traffic_bytes, l2_hit}}; empty when no pass exists."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return {}
    out = {}
    # C5's kernels, width and batch are those of the HBM-resident leg ("C5hbm": the passes wrap tools/run_c5_shape.py)
    for name in (config, config + "hbm"):
        for e in entries:
            if (e.get("config"), e.get("k"), e.get("batch_rows")) == (name, k, batch_rows):
                out.setdefault(e["kernel"], e)
    return out


STEP_KERNELS = ("k_forward", "k_backward", "k_fixup", "k_apply")
PMC_STATE = {"dead": False}      # a counter pass that had to be killed ends the live collection for the run


def pmc_pass(counters, child_argv, skip=4, timeout_s=150, per_step=None):
    """One `rocprofv3 --pmc <counters> -- python3 <child_argv>` run (counter collection only: no trace domain beside it) as
    a CHILD process; -> {kernel: {counter: mean per dispatch after the first `skip` dispatches of that kernel}} for the
    kernels of the SGD step, or None when rocprofv3 is not there / fails (the caller falls back to the committed profile).
    per_step = (warmup_steps, timed_steps) of the child: a kernel the step launches several times (the data-parallel step's
    backward runs once per feature interval) is then summed over a step's launches — the figure is per STEP of that kernel.
    Kernel names are folded as in tools/make_pmc_json.py (k_forward_wt -> k_forward, k_backward_p -> k_backward, ...).
    A pass that runs into its time limit is killed with its whole process group (rocprofv3's grandchild would otherwise keep
    the GPU busy beside the timed legs that follow) and ends the live collection for this run."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out_dir = tempfile.mkdtemp(prefix="fmhip_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + list(counters) + ["-d", out_dir, "-o", "pmc", "--output-format", "csv", "--", sys.executable] + list(child_argv)
        env = dict(os.environ, TMPDIR="/tmp")
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            PMC_STATE["dead"] = True
            sys.stderr.write("[bench] rocprofv3 --pmc %s ran into its %d s limit: process group killed, no further counter passes in this run\n" %
                             (" ".join(counters), timeout_s))
            return None
        if proc.returncode != 0:
            sys.stderr.write("[bench] rocprofv3 --pmc %s failed (rc %d): %s\n" % (" ".join(counters), proc.returncode, err.decode()[-400:]))
            return None
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"])
                if not m or not m.group(1).startswith(STEP_KERNELS):
                    continue
                kn = m.group(1).replace("k_forward_wt", "k_forward").replace("k_forward_lds", "k_forward").replace("k_backward_p", "k_backward").replace("k_apply_rows", "k_apply")
                agg[kn][row["Counter_Name"]].append(float(row["Counter_Value"]))
        def mean(v):
            if per_step and len(v) % (per_step[0] + per_step[1]) == 0:
                lps = len(v) // (per_step[0] + per_step[1])              # launches of this kernel per step
                return sum(v[per_step[0] * lps:]) / per_step[1]
            return sum(v[skip:]) / max(len(v[skip:]), 1)
        return {kn: {cn: mean(v) for cn, v in cs.items() if len(v) > skip} for kn, cs in agg.items()} or None
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as ex:
        sys.stderr.write("[bench] rocprofv3 --pmc pass failed: %r\n" % (ex,))
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def live_pmc(child_argv, per_step=None):
    """Fabric-side traffic per launch of the step's kernels, measured NOW: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
    do not fit one) over tools/pmc_leg.py running the same workload.  Units and the gfx950 correction as
    MI355X_MICROARCH.md prescribes: both counters are KiB; FETCH_SIZE tallies the 128-B requests of wide (16 B per lane)
    reads at 64 B — the row gathers and the dense block's streams are such reads, the 4-B index / value streams are not and
    the counter cannot tell them apart, so the doubled figure is an upper bound.  -> {kernel: {...}} or None."""
    if PMC_STATE["dead"]:
        return None
    fetch = pmc_pass(["FETCH_SIZE"], child_argv, per_step=per_step)
    write = pmc_pass(["WRITE_SIZE"], child_argv, per_step=per_step) if fetch and not PMC_STATE["dead"] else None
    if not fetch or not write:
        return None
    out = {}
    for kn in fetch:
        fr, wr = fetch[kn].get("FETCH_SIZE"), write.get(kn, {}).get("WRITE_SIZE")
        if fr is None or wr is None:
            continue
        out[kn] = {"fetch_raw_bytes": int(fr * 1024), "fetch_corrected_bytes": int(2 * fr * 1024), "write_bytes": int(wr * 1024),
                   "traffic_bytes": int(2 * fr * 1024 + wr * 1024)}
    if out:
        out["step"] = {"traffic_bytes": sum(e["traffic_bytes"] for e in out.values())}
        # the hit rates of THIS run (one more pass: the L2's hits / misses and the L1s' accesses / requests passed on to L2 fit
        # one counter set): what the gather ceilings are blended with, instead of the committed profile's figure
        hits = None if PMC_STATE["dead"] else pmc_pass(["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
                                                       child_argv, per_step=per_step)
        for kn, c in (hits or {}).items():
            if kn in out and c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
                out[kn]["l2_hit"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
                out[kn]["l2_hit_measured_in_this_run"] = True
                if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                    out[kn]["l1_hit_share_of_accesses"] = 1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    return out or None

