"""FMUtils — libFM text I/O, host mirror of S/fm/FMUtils.scala:23-69 (the on-disk format either
side of the training path).  Pure host code; rows end up in a DataSet (CSR) ready for cache()."""
from decimal import ROUND_HALF_EVEN, Decimal

import numpy as np

from .dataset import DataSet


def loadLibFMFile(path, numFeatures=-1, **dataset_kw):
    """S/fm/FMUtils.scala:23-56.  Lines are trimmed; empty lines and lines starting with '#' are
    skipped; `label idx:val idx:val ...` split on single spaces, empty items ignored.  Indices are
    kept AS WRITTEN (0-based; the reference does not shift them — quirk Q9) and need not be sorted.
    The vector length is numFeatures + 1, or (max index) + 1 when numFeatures <= 0 (:42-53); it only
    matters through DataSet.dimension, which is data-driven here as in S/DataSet.scala:27-29."""
    ys, cols, vals, ptr = [], [], [], [0]
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            items = line.split(" ")
            ys.append(float(items[0]))
            n = 0
            for item in items[1:]:
                if not item:
                    continue
                i, v = item.split(":")
                cols.append(int(i))
                vals.append(float(v))
                n += 1
            ptr.append(ptr[-1] + n)
    col = np.asarray(cols, np.int32)
    if numFeatures > 0 and len(col) and int(col.max()) > numFeatures:
        raise ValueError("feature index %d exceeds numFeatures = %d" % (int(col.max()), numFeatures))
    return DataSet(np.asarray(ptr, np.int64), col, np.asarray(vals, np.float64), np.asarray(ys, np.float64),
                   **dataset_kw)


def minimizeString(v):
    """S/fm/FMUtils.scala:71-74: java.text.DecimalFormat("#") for integral values, "#.###" otherwise
    (HALF_EVEN, no trailing zeros, and — DecimalFormat semantics — no leading zero: 0.5 -> ".5")."""
    v = float(v)
    if v == np.floor(v) and not np.isinf(v):
        s = "%d" % int(v)
        return "-0" if (v == 0 and np.signbit(v)) else s
    if np.isnan(v):
        return "�"                       # DecimalFormat's NaN symbol
    if np.isinf(v):
        return "∞" if v > 0 else "-∞"
    d = Decimal(repr(v)).quantize(Decimal("0.001"), rounding=ROUND_HALF_EVEN)
    neg = d < 0
    s = format(abs(d), "f").rstrip("0").rstrip(".")
    if s.startswith("0."):
        s = s[1:]
    if s in ("", "0"):
        return "-0" if neg else "0"
    return ("-" if neg else "") + s


def saveAsLibFMFile(dataset, path, index_offset=1):
    """S/fm/FMUtils.scala:58-69: `label i+1:value ...` — the reference writes indices shifted by +1
    (index_offset = 1) although its loader reads them unshifted (quirk Q9); pass index_offset=0 for
    files that round-trip through loadLibFMFile."""
    with open(path, "w") as fh:
        for label, (idx, val) in dataset.rows():
            fh.write(minimizeString(label))
            for i, v in zip(idx, val):
                fh.write(" %d:%s" % (int(i) + index_offset, minimizeString(v)))
            fh.write("\n")
