#!/usr/bin/env python3
"""Does the backward's time move in ROUNDS of resident waves?  (profiles/r04_experiments.md section 21)

A slot of the column walk takes one 64-entry range; a wave holds 8 (Kp = 32) or 4 (Kp = 64) slots; the chip holds
256 CUs x 4 SIMDs x W waves.  If the launch ran as rounds of resident waves, its time per entry would jump where the
wave count crosses a multiple of the resident count; if the dispatcher keeps the chip full until the end it is flat.
One dataset, a sweep of batch sizes: us per launch and ns per 1000 sparse entries, forward / backward / fixup.

python3 tools/batch_quant_probe.py [C3|C2] [first last step (rows, default 200000 340000 10000)]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
first, last, step = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (200_000, 340_000, 10_000)
cfg = synth.CONFIGS[cfg_name]
d = synth.make_config(cfg_name, rows=1_000_000)
L = _ffi.load()
print("# %s k=%d; waves resident at W=4: %d" % (cfg_name, cfg["k"], 256 * 4 * 4), flush=True)
print("# batch_rows  sparse_bwd_per_batch  ranges_per_batch  waves  rounds@4  fwd_us  bwd_us  fix_us  bwd_ns_per_1k_entries", flush=True)
for rows in range(first, last + 1, step):
    ds = DataSet.from_arrays(d, batch_rows=rows).cache()
    nb = ds.n_batches
    full = [b for b in range(nb) if ds.batch_info(b)["rows"] == rows] or [0]       # the ragged last batch stays out
    lay = ds.layout()
    fm = FMModel(cfg["features"] - 1, cfg["k"], seed=3, init_on_device=True)
    hm, hd = fm.handle, ds.handle
    for j in range(6):
        _ffi.check(L.fmhip_sgd_step(hm, hd, full[j % len(full)], 0.02, 0.0, 1e-4, 1e-4, None))
    _ffi.check(L.fmhip_profile_begin(hm))
    n = 60
    for j in range(n):
        _ffi.check(L.fmhip_sgd_step(hm, hd, full[j % len(full)], 0.02, 0.0, 1e-4, 1e-4, None))
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
    us = [p.ms[i] / max(p.launches[i], 1) * 1e3 for i in range(4)]
    # the dataset's totals cover every batch (the ragged one too): scale to one full batch by rows
    total_rows = sum(ds.batch_info(b)["rows"] for b in range(nb))
    sp = lay["nnz_sparse_backward"] * rows / total_rows
    rg = lay["ranges"] * rows / total_rows
    slots_per_wave = 64 // (8 if cfg["k"] <= 32 else 16)
    waves = rg / slots_per_wave
    print("%9d  %12.0f  %10.0f  %8.0f  %6.2f  %7.1f  %7.1f  %6.1f  %8.2f" % (rows, sp, rg, waves, waves / (256 * 16), us[0], us[2], us[3],
                                                                            us[2] * 1e6 / max(sp, 1.0)), flush=True)
    fm.close()
    ds.unpersist()
