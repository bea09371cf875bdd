"""Shared test helpers: KAT decoding and small random CSR problems."""
import numpy as np


def f(x):
    """KAT json node -> float / nested float lists."""
    if isinstance(x, dict) and "f" in x:
        return x["f"]
    if isinstance(x, list):
        return [f(v) for v in x]
    return x


def kat_arrays(case):
    """-> dict with w0, w (n1,), v (k,n1), row_ptr, col, val, y as numpy arrays."""
    rows = case["rows"]
    row_ptr = np.zeros(len(rows) + 1, np.int64)
    col, val = [], []
    for r, row in enumerate(rows):
        for i, x in row:
            col.append(i)
            val.append(f(x))
        row_ptr[r + 1] = len(col)
    return dict(k=case["k"], n1=case["n1"], w0=f(case["w0"]), w=np.array(f(case["w"]), np.float64),
                v=np.array(f(case["V"]), np.float64).reshape(case["k"], case["n1"]),
                row_ptr=row_ptr, col=np.array(col, np.int32), val=np.array(val, np.float64),
                y=np.array(f(case["y"]), np.float64))


def random_problem(seed, n_rows, n1, k, nnz_lo, nnz_hi, empty_rows=(), scale=0.1, sort_idx=False):
    """Random CSR rows with distinct (unsorted unless sort_idx) indices per row."""
    rng = np.random.default_rng(seed)
    row_ptr = np.zeros(n_rows + 1, np.int64)
    col, val = [], []
    for r in range(n_rows):
        nnz = 0 if r in empty_rows else int(rng.integers(nnz_lo, nnz_hi + 1))
        nnz = min(nnz, n1)
        idx = rng.choice(n1, size=nnz, replace=False).astype(np.int32)
        if sort_idx:
            idx.sort()
        x = np.where(rng.random(nnz) < 0.5, 1.0, rng.uniform(0.1, 1.0, nnz))
        col.append(idx)
        val.append(x)
        row_ptr[r + 1] = row_ptr[r] + nnz
    col = np.concatenate(col) if col else np.zeros(0, np.int32)
    val = np.concatenate(val) if val else np.zeros(0)
    # make the last slot appear so that dimension == n1 - 1
    if len(col) and not (col == n1 - 1).any():
        col[0] = n1 - 1 if (col[row_ptr[0]:row_ptr[1]] != n1 - 1).all() else col[0]
    w0 = float(rng.normal(0, scale))
    w = rng.normal(0, scale, n1)
    v = rng.normal(0, scale, (k, n1))
    y = rng.normal(0, 1.0, n_rows)
    return dict(k=k, n1=n1, w0=w0, w=w, v=v, row_ptr=row_ptr, col=col.astype(np.int32),
                val=val.astype(np.float64), y=y)


def build_jni_harness(tmp_path, sanitize=False):
    """Compiles jvm/fmhip_jni.c (the JNI shim: source-only, the image has no JDK) together with tests/jni_harness.c against
    the stand-in tests/jni_stub/jni.h, warnings as errors — every call of the shim into include/fmhip.h is type-checked — and
    returns the executable (`host` / `gpu`: tests/jni_harness.c)."""
    import os
    import subprocess
    from sparkfm_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / ("jni_harness_asan" if sanitize else "jni_harness"))
    # sanitize: AddressSanitizer + UBSan over the shim and the harness (CPU build only; libfmhip.so itself is not instrumented)
    san = ["-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else []
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror"] + san + [
                           "-I" + os.path.join(root, "tests", "jni_stub"), "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "jni_harness.c"), os.path.join(root, "jvm", "fmhip_jni.c"),
                           "-L" + _build.LIBDIR, "-lfmhip", "-Wl,-rpath," + _build.LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def build_cpp_mirror(tmp_path):
    """tests/cpp_mirror.cpp — a consumer of include/sparkfm.hpp (header-only C++ mirror of the reference's host classes over the
    product C ABI) — compiled as strict C++17 with warnings as errors and linked against libfmhip.so; -> the executable."""
    import os
    import subprocess
    from sparkfm_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cpp_mirror")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp_mirror.cpp"), "-L" + _build.LIBDIR, "-lfmhip",
                           "-Wl,-rpath," + _build.LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe
