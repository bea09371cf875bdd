"""Byte accounting, ceilings and the `roofline` block of bench.py's JSON line (pure arithmetic: importable without a GPU)."""
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); measured streaming copy ~6.3e12
# Ceilings for gathers of whole 128-B-multiple rows by where the table lives (MI355X_MICROARCH.md,
# "Indexed rows"): the XCD's own L2, the Infinity Cache, HBM (measured sweep / spec peak)
# (the UPPER end of the guide's measured ranges — 16.8-18.8 TB/s from L2 — so that a "ceiling" is one: a kernel that also
# hits in its CU's L1, which these rates do not price, must not pass it)
CEIL = {"l2_gather": 18.8e12, "mall_gather": 8.6e12, "hbm_gather": 6.1e12, "hbm_stream": 6.3e12}
L2_BYTES_PER_XCD = 4 << 20
MALL_BYTES = 256 << 20


def alg_bytes(k):
    """SURVEY.md §8(d) algorithmic bytes per stored nonzero, split by kernel (fp32/int32):
    forward  = col 4 + val 4 + V-row read 4k + w read 4      = 4k + 12
    backward = V-grad row add 4k + w-grad add 4               = 4k + 4   (SURVEY's figure; the walk itself reads 8 + 4k per entry)
    whole step B_alg(k) = 8k + 16 (plus 16 B/row and 12(n+1)(k+1) B/step for the dense update)."""
    return {"forward": 4 * k + 12, "backward": 4 * k + 4, "step": 8 * k + 16}


def requested_bytes(kp, rows, nnz, nnz_sparse, n_cols, hot, touched_rows, dense_apply, n1p, packed, nnz_sparse_bwd=None, hot_pages=1):
    """Bytes each kernel of one step actually ASKS the memory system for (our own count of its loads and
    stores, whatever level serves them), and the table its gathers hit.  nnz_sparse / nnz_sparse_bwd: the entries
    of the batch in the CSR stream (forward) / in the transposed stream (backward: fewer, the gradient-side pages
    of the dense hot block are not in it); the block product streams P once and 64 B per row and page."""
    row = 4 * kp
    hot_b = 64 * rows if hot else 0
    if nnz_sparse_bwd is None:
        nnz_sparse_bwd = nnz_sparse
    fwd = nnz_sparse * (8 + row + (0 if packed else 4)) + rows * (8 + 4 + row + 4) + hot_b
    # (no separate residual read: with a spare slot e sits in the P row, without one it rides in the row's low mantissa bits)
    bwd = nnz_sparse_bwd * (8 + row) + n_cols * (row + 8) + (rows * row + hot_b * max(hot_pages, 1) if hot else 0)
    apply_rows = n1p if dense_apply else touched_rows
    app = apply_rows * (3 * row + 16)          # V read+write, G read (+ zero store counted with the write)
    return {"forward": fwd, "backward": bwd, "apply": app}


def gather_ceiling(table_bytes, l2_hit=None):
    """Ceiling for a kernel bound by gathers from a table of `table_bytes` that every XCD reads: the
    blend of the L2 and Infinity-Cache gather rates at L2 hit rate h (measured by rocprofv3 where a
    committed profile exists, else the uniform-gather share min(1, 4 MiB / table)); tables beyond the
    Infinity Cache gather at the HBM rate."""
    if table_bytes > MALL_BYTES:
        if l2_hit is None:
            return "hbm_gather", CEIL["hbm_gather"], None
        # skewed gathers from a table in HBM: the measured share hits L2; what misses is served by the Infinity Cache or by
        # HBM in a proportion no counter separates — priced at the faster of the two, so this stays an upper bound
        c = 1.0 / (l2_hit / CEIL["l2_gather"] + (1.0 - l2_hit) / CEIL["mall_gather"])
        return "l2_gather x %.2f + mall_gather x %.2f (L2 misses priced at the Infinity-Cache rate: upper bound)" % (l2_hit, 1.0 - l2_hit), c, l2_hit
    h = l2_hit if l2_hit is not None else min(1.0, L2_BYTES_PER_XCD / max(table_bytes, 1))
    c = 1.0 / (h / CEIL["l2_gather"] + (1.0 - h) / CEIL["mall_gather"])
    return "l2_gather x %.2f + mall_gather x %.2f" % (h, 1.0 - h), c, h


def roofline_block(kern, dom, ab, pd, pmc, step_ms, live):
    """`roofline` of the JSON line for the dominant kernel `dom`: achieved = bytes the rocprofv3 counters saw leave the L2s per
    launch (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md's units and gfx950 correction) / that kernel's launch duration
    measured by HIP events in THIS run; frac = achieved / 8 TB/s.  The algorithmic figure of SURVEY §8(d) is kept beside it,
    flagged: it prices every stored nonzero at a gathered row and is not an HBM rate."""
    e = kern.get(dom, {})
    avg_ms = e.get("avg_ms")
    traffic = e.get("traffic_bytes")
    basis = "counters"
    if traffic is None:           # no profile of this configuration anywhere: our own count of the kernel's loads and stores
        traffic, basis = e.get("requested_bytes_per_launch"), "requested bytes (no counter pass exists for this configuration)"
    achieved = traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms else None
    requested_only = None
    if basis != "counters":
        # our own count of the kernel's loads and stores is what it ASKS of the memory system, caches included — not an HBM-side
        # figure and no roofline: reported beside the (empty) roofline, never as its `achieved`
        requested_only = {"requested_bytes": traffic, "requested_GBps": achieved, "note": "no counter pass exists for this configuration: no HBM-side figure is claimed"}
        achieved = None
    step_traffic = pmc.get("step", {}).get("traffic_bytes")
    alg = e.get("alg_GBps")
    return {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved * 1e9 / HBM_PEAK if achieved else None, "traffic": traffic if basis == "counters" else None, "basis": basis,
            "requested_only": requested_only,
            "traffic_source": e.get("traffic_source"), "traffic_measured_in_this_run": bool(live),
            "avg_launch_ms": avg_ms, "nnz_per_launch": pd[dom]["nnz"] / max(pd[dom].get("steps") or pd[dom]["launches"], 1),
            "launches_per_step": e.get("launches_per_step", 1),
            "what": "achieved = fabric-side bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: what left the L2s; Infinity-Cache hits "
                    "are included — the part exposes no DRAM-side or MALL hit counter, profiles/README.md — so an upper bound on HBM bytes) / "
                    "the launch's duration by HIP events in this run; frac = achieved / 8 TB/s",
            "step": {"traffic": step_traffic, "achieved": step_traffic / (step_ms * 1e-3) / 1e9 if step_traffic else None,
                     "frac": step_traffic / (step_ms * 1e-3) / HBM_PEAK if step_traffic else None,
                     "what": "the same for the whole step: the counters' bytes of all its kernels / the measured step time"},
            "algorithmic_bytes_per_nnz": ab[dom], "algorithmic_achieved": alg, "algorithmic_frac": alg * 1e9 / HBM_PEAK if alg else None,
            "algorithmic_note": "SURVEY §8(d)'s figure (4k+4 B for EVERY stored nonzero of the batch / launch time): NOT an HBM rate and may "
                                "pass the peak — V, P and the gradient live in L2 / Infinity Cache at this size and the nonzeros of the dense "
                                "hot block cost one streamed value instead of a gathered row",
            "requested_GBps": e.get("requested_GBps"), "ceiling": e.get("ceiling"), "frac_of_ceiling": e.get("frac_of_ceiling")}


def compulsory_hbm_bytes(kp, rows, nnz_sparse_fwd, nnz_sparse_bwd, n_cols, n1p, hot_pages):
    """Bytes ONE step cannot avoid moving through HBM however well the caches work (the floor under the step, not what it
    moves): every index / value stream once (the dataset does not fit any cache across the steps of an epoch), the dense hot
    block's pages once, and one pass each over P (rows x Kp), the parameter table V and the packed gradient G.  Gathered
    rows are NOT priced per nonzero here — at C2-C4 V, P and G live in L2 / Infinity Cache and every re-read is a cache hit."""
    streams = 8 * nnz_sparse_fwd + 12 * rows + 8 * nnz_sparse_bwd + 8 * n_cols + 4 * (nnz_sparse_bwd // 64 + 1)
    xhot = 64 * rows * max(hot_pages, 0)
    tables = 4 * kp * rows + 4 * (kp + 2) * n1p * 2            # P; V (+ w) and G (+ G_w, G_b)
    return {"streams": streams, "xhot": xhot, "P_V_G_one_pass_each": tables, "total": streams + xhot + tables}


def annotate_roofline(roof, compulsory, step_ms, step_ceiling_frac=None):
    """What the block says about itself: the compulsory HBM bytes of a step and the time they take at the
    peak (the step's HBM floor), the step's fraction of its kernels' own gather ceilings, and — when SURVEY section 8(d)'s
    algorithmic rate passes the peak — the label that says why that is not a bug."""
    roof["compulsory_hbm_bytes_per_step"] = compulsory["total"]
    roof["compulsory_hbm_bytes_parts"] = {k: v for k, v in compulsory.items() if k != "total"}
    roof["hbm_floor_ms"] = compulsory["total"] / HBM_PEAK * 1e3
    roof["hbm_floor_share_of_step"] = roof["hbm_floor_ms"] / step_ms if step_ms else None
    if step_ceiling_frac is not None:
        roof["step_ceiling_frac"] = step_ceiling_frac
        roof["step_ceiling_note"] = ("sum over the step's kernels of (bytes requested / the ceiling that binds the kernel: L2 and Infinity-Cache "
                                     "gather rates blended by the measured L2 hit rate, the HBM stream rate for the update) / measured step time")
    af = roof.get("algorithmic_frac")
    if af is not None and af > 1.0:
        roof["algorithmic_label"] = ("model not a bound: tables cache-resident, hot block gathers nothing — the step is bound by cache / texture-path "
                                     "latency (see frac_of_ceiling, step_ceiling_frac), not by HBM; its HBM floor is hbm_floor_ms")
    return roof


def kernel_table(prof, k, kp, req, pmc, table_bytes):
    """Per-kernel: HIP-event time, algorithmic rate, requested-byte rate, the ceiling
    that binds it and the fraction of THAT ceiling.  A kernel the data-parallel step launches once per feature interval
    (backward, fixup, update) is summed over the launches of one step — the profile counts the steps each kind was timed
    in — so its time, nonzeros and bytes are per STEP, which is what the requested-byte and counter figures beside them are."""
    ab = alg_bytes(k)
    pd = prof.as_dict()
    tot_ms = max(sum(x["ms"] for x in pd.values()), 1e-12)
    kern = {}
    for name, p in pd.items():
        if not p["launches"]:
            continue
        steps = max(p.get("steps") or p["launches"], 1)
        lps = p["launches"] / steps
        avg_ms = p["ms"] / steps
        ent = {"avg_ms": avg_ms, "launches": p["launches"], "share": p["ms"] / tot_ms}
        if lps > 1:
            ent["launches_per_step"] = lps
            ent["avg_ms_is"] = "the sum over the %.3g launches of one step (one per feature interval)" % lps
        if name in ab:
            ent["alg_bytes_per_nnz"] = ab[name]
            ent["alg_GBps"] = (p["nnz"] / steps) * ab[name] / (avg_ms * 1e-3) / 1e9
        if name in req:
            ent["requested_bytes_per_launch"] = req[name]
            ent["requested_GBps"] = req[name] / (avg_ms * 1e-3) / 1e9
            if name in ("forward", "backward"):
                hit = pmc.get("k_" + name, {}).get("l2_hit")
                cname, c, h = gather_ceiling(table_bytes[name], hit)
                if cname == "hbm_gather":
                    # a table beyond the Infinity Cache and no measured hit rate: skewed gathers are served by the caches in a
                    # share nobody measured here, so no rate is a ceiling for them — none is claimed
                    ent["ceiling"] = None
                    ent["frac_of_ceiling"] = None
                    ent["ceiling_note"] = "no ceiling claimed: the table is beyond the Infinity Cache and this configuration has no measured L2 hit rate"
                else:
                    measured = pmc.get("k_" + name, {}).get("l2_hit_measured_in_this_run")
                    ent["ceiling"] = {"name": cname, "GBps": c / 1e9, "table_bytes": table_bytes[name], "l2_hit": h,
                                      "l2_hit_source": (("rocprofv3 --pmc TCC_HIT / TCC_MISS pass of this run" if measured else "profiles/pmc_traffic.json")
                                                        if hit is not None else ("uniform-gather model" if h is not None else None))}
                    l1 = pmc.get("k_" + name, {}).get("l1_hit_share_of_accesses")
                    if l1 is not None:
                        ent["ceiling"]["l1_hit_share_of_accesses"] = l1      # served by the CU's own L1: not priced by the ceiling (it only adds headroom)
            else:
                ent["ceiling"] = {"name": "hbm_stream", "GBps": CEIL["hbm_stream"] / 1e9}
            if ent.get("ceiling"):
                ent["frac_of_ceiling"] = ent["requested_GBps"] / ent["ceiling"]["GBps"]
            if (ent.get("frac_of_ceiling") or 0.0) > 1.0:
                ent["ceiling_exceeded"] = ("the kernel asked for bytes faster than the L2 / Infinity-Cache gather rates allow: the excess was "
                                           "served by the CUs' L1s, which the ceiling does not price")
        pe = pmc.get("k_" + name, {})
        if pe.get("traffic_bytes") is not None:
            ent["traffic_bytes"] = pe["traffic_bytes"]           # fabric-side: FETCH_SIZE x2 + WRITE_SIZE per launch
            ent["traffic_source"] = pe.get("traffic_source", "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this configuration)")
            ent["traffic_GBps"] = pe["traffic_bytes"] / (avg_ms * 1e-3) / 1e9
            ent["traffic_frac_of_8TBps"] = ent["traffic_GBps"] * 1e9 / HBM_PEAK
        kern[name] = ent
    if "apply" in req and "apply" not in kern and "fixup" in kern:
        # merged finish: the dense update ran inside the fixup launch (fmhip_tune key 11)
        ent = kern["fixup"]
        ent["includes"] = "the parameter update (merged finish)"
        ent["requested_bytes_per_launch"] = req["apply"]
        ent["requested_GBps"] = req["apply"] / (ent["avg_ms"] * 1e-3) / 1e9
        ent["ceiling"] = {"name": "hbm_stream", "GBps": CEIL["hbm_stream"] / 1e9}
        ent["frac_of_ceiling"] = ent["requested_GBps"] / ent["ceiling"]["GBps"]
    return kern

