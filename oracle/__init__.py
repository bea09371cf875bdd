"""CPU oracle for the FM hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``sparkfm_amd``) never does.

Parity status: **parity unpinned by the reference** — SparkFM ships no tests or
golden vectors and cannot be built here (no JVM).  The oracle is pinned by the
exact-rational known-answer vectors in ``tests/golden/`` instead.
"""
from .capi import (  # noqa: F401
    build,
    lib,
    predict,
    rmse,
    residual,
    transpose,
    dimension,
    term_q,
    batch_grad,
    sgd_step,
    sgd_epoch,
    als_epoch,
    max_threads,
)
