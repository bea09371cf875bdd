#!/usr/bin/env python3
"""Generates tests/golden/kat_fm.json — exact-rational known-answer vectors.

The reference (SparkFM) has no tests and no golden vectors, and it
cannot be run here (no JVM), so these vectors are derived INDEPENDENTLY of both the
reference's rearranged formula and of oracle/fm_oracle.c:

* yhat comes from the NAIVE pairwise definition
      yhat = w0 + sum_i w_i x_i + sum_{i<j} <v_i, v_j> x_i x_j
  (not from 0.5*sum_f[(sum v x)^2 - sum (v x)^2], S/fm/FMModel.scala:50), so the
  rearrangement itself is cross-checked;
* d yhat / d v_{f,i} = x_i * sum_{j != i} v_{f,j} x_j  (differentiating the naive form);
* everything is computed with fractions.Fraction, then stored both as "p/q"
  strings and as the nearest float.

The ALS epoch section restates S/fm/lib/ALS.scala:15-75,152-198 in 120-digit
decimal arithmetic (exact rationals blow up: every closed-form step squares the
denominators), including quirk Q1 (the loop `0 until num_attribute` never trains
the last slot).  120 digits is ~100 digits beyond fp64, i.e. exact for the
purpose of checking an fp64 restatement.

Run:  python tests/golden/make_kats.py   (writes kat_fm.json next to itself)
"""
import json
import os
import random
from decimal import Decimal, getcontext
from fractions import Fraction as F

getcontext().prec = 120


def D(x):
    """Fraction -> 120-digit Decimal."""
    return Decimal(x.numerator) / Decimal(x.denominator)


def naive_predict(w0, w, V, row):
    """row: list of (idx, val).  V[f][i]."""
    k = len(V)
    y = w0
    for i, x in row:
        y += w[i] * x
    for a in range(len(row)):
        for b in range(a + 1, len(row)):
            ia, xa = row[a]
            ib, xb = row[b]
            dot = sum(V[f][ia] * V[f][ib] for f in range(k))
            y += dot * xa * xb
    return y


def q_of(V, row):
    return [sum(V[f][i] * x for i, x in row) for f in range(len(V))]


def dyhat_dv(V, row):
    """{idx: [d/dv_{f,idx} for f]} from the naive form."""
    out = {}
    for i, x in row:
        out[i] = [x * sum(V[f][j] * xj for j, xj in row if j != i) for f in range(len(V))]
    return out


def sgd_step(w0, w, V, rows, ys, eta, reg0, regw, regv):
    """theta <- theta - eta*(g/|B| + lambda*theta); g = sum_r e_r * dyhat_r/dtheta."""
    k, n1, B = len(V), len(w), len(rows)
    g0 = F(0)
    gw = [F(0)] * n1
    gV = [[F(0)] * n1 for _ in range(k)]
    for row, y in zip(rows, ys):
        e = naive_predict(w0, w, V, row) - y
        g0 += e
        d = dyhat_dv(V, row)
        for i, x in row:
            gw[i] += e * x
            for f in range(k):
                gV[f][i] += e * d[i][f]
    w0n = w0 - eta * (g0 / B + reg0 * w0)
    wn = [w[i] - eta * (gw[i] / B + regw * w[i]) for i in range(n1)]
    Vn = [[V[f][i] - eta * (gV[f][i] / B + regv * V[f][i]) for i in range(n1)] for f in range(k)]
    return w0n, wn, Vn, g0, gw, gV


def als_epoch(w0, w, V, rows, ys, reg0, regw, regv):
    """S/fm/lib/ALS.scala:15-75 in 120-digit decimals.  num_attribute = len(w) - 1."""
    F = lambda x: Decimal(x)                       # shadows Fraction inside this function
    w0, reg0, regw, regv = D(w0), D(reg0), D(regw), D(regv)
    w = [D(x) for x in w]
    V = [[D(x) for x in r] for r in V]
    rows = [[(i, D(x)) for i, x in r] for r in rows]
    ys = [D(y) for y in ys]
    k, n1 = len(V), len(w)
    num_attribute = n1 - 1
    N = len(rows)
    w = list(w)
    V = [list(r) for r in V]

    def upd(nv, ov):                       # :190-192 (rationals are never NaN/Inf)
        return nv is not None and nv != ov

    def theta(th, reg, seh, shs):          # :167-176
        den = reg + shs
        nv = None if den == 0 else -(seh - th * shs) / den
        return nv if upd(nv, th) else th

    e = [naive_predict(w0, w, V, rows[r]) - ys[r] for r in range(N)]      # :17,142-144
    w0n = theta(w0, reg0, sum(e, F(0)), F(N))                             # :21,152-154
    if upd(w0n, w0):
        e = [er + (w0n - w0) for er in e]                                  # :24
    w0 = w0n
    cols = {}
    for r, row in enumerate(rows):                                         # transpose S/DataSet.scala:31-38
        for i, x in row:
            cols.setdefault(i, []).append((r, x))
    for i in range(num_attribute):                                         # :38 (Q1: slot n skipped)
        if i not in cols:
            continue
        h = cols[i]
        shs = sum((x * x for _, x in h), F(0))
        seh = sum((e[r] * x for r, x in h), F(0))
        nv = theta(w[i], regw, seh, shs)
        if upd(nv, w[i]):
            for r, x in h:
                e[r] += x * (nv - w[i])
        w[i] = nv
    for f in range(k):                                                     # :47
        q = [F(0)] * N
        for i, c in cols.items():                                          # :50,146-150 (all slots incl. n)
            for r, x in c:
                q[r] += V[f][i] * x
        for i in range(num_attribute):                                     # :52
            if i not in cols:
                continue
            vfi = V[f][i]
            h = [(r, x * q[r] - x * x * vfi) for r, x in cols[i]]          # :56-58
            shs = sum((hv * hv for _, hv in h), F(0))
            seh = sum((e[r] * hv for r, hv in h), F(0))
            nv = theta(vfi, regv, seh, shs)
            if upd(nv, vfi):
                for r, hv in h:
                    e[r] += hv * (nv - vfi)
            for r, x in cols[i]:                                           # :60-62
                q[r] += x * (nv - vfi)
            V[f][i] = nv                                                   # :64
    return w0, w, V


def ser(x):
    if isinstance(x, F):
        # the exact "p/q" string is kept only while it stays readable; the float is always
        # the correctly rounded value of the exact rational
        if x.numerator.bit_length() + x.denominator.bit_length() > 256:
            return {"f": float(x)}
        return {"q": "%d/%d" % (x.numerator, x.denominator), "f": float(x)}
    if isinstance(x, Decimal):
        return {"f": float(x)}
    if isinstance(x, (list, tuple)):
        return [ser(v) for v in x]
    if isinstance(x, dict):
        return {str(k): ser(v) for k, v in x.items()}
    return x


def make_case(name, w0, w, V, rows, ys, eta, reg0, regw, regv, als_regs):
    k, n1 = len(V), len(w)
    yh = [naive_predict(w0, w, V, r) for r in rows]
    e = [a - b for a, b in zip(yh, ys)]
    sse = sum((x * x for x in e), F(0))
    w0n, wn, Vn, g0, gw, gV = sgd_step(w0, w, V, rows, ys, eta, reg0, regw, regv)
    a0, aw, aV = als_epoch(w0, w, V, rows, ys, *als_regs)
    return {
        "name": name, "k": k, "n1": n1,
        "w0": ser(w0), "w": ser(w), "V": ser(V),
        "rows": [[[i, ser(x)] for i, x in r] for r in rows], "y": ser(ys),
        "yhat": ser(yh), "e": ser(e), "q": ser([q_of(V, r) for r in rows]),
        "dyhat_dv": [ser(dyhat_dv(V, r)) for r in rows],
        "sse": ser(sse), "mse": ser(sse / len(rows)),
        "grad": {"g0": ser(g0), "gw": ser(gw), "gV": ser(gV)},
        "sgd": {"eta": ser(eta), "reg0": ser(reg0), "regw": ser(regw), "regv": ser(regv),
                "w0": ser(w0n), "w": ser(wn), "V": ser(Vn)},
        "als": {"reg0": ser(als_regs[0]), "regw": ser(als_regs[1]), "regv": ser(als_regs[2]),
                "w0": ser(a0), "w": ser(aw), "V": ser(aV)},
    }


def survey_case():
    """The vector printed in SURVEY.md §4."""
    w0 = F(1, 2)
    w = [F(1, 10), F(-1, 5), F(3, 10), F(0)]
    V = [[F(1, 10), F(1, 5), F(-3, 10), F(2, 5)], [F(-1, 2), F(3, 5), F(7, 10), F(-4, 5)]]
    rows = [[(0, F(1)), (2, F(2)), (3, F(1, 2))], [(1, F(1)), (2, F(1))], [(0, F(3))]]
    ys = [F(1), F(-1), F(2)]
    return make_case("survey_s4", w0, w, V, rows, ys, F(1, 10), F(0), F(0), F(0), (F(0), F(0), F(10)))


def random_case(name, seed, k, n1, n_rows, max_nnz, empty_row=False, unsorted=True):
    rng = random.Random(seed)
    fr = lambda lo, hi, den: F(rng.randint(lo, hi), den)
    w0 = fr(-10, 10, 10)
    w = [fr(-10, 10, 20) for _ in range(n1)]
    V = [[fr(-10, 10, 10) for _ in range(n1)] for _ in range(k)]
    rows, ys = [], []
    for r in range(n_rows):
        nnz = rng.randint(1, max_nnz)
        if empty_row and r == 1:
            nnz = 0
        idx = rng.sample(range(n1), min(nnz, n1))
        if not unsorted:
            idx.sort()
        rows.append([(i, F(1) if rng.random() < 0.5 else fr(1, 10, 10)) for i in idx])
        ys.append(fr(-20, 20, 10))
    # make sure the last slot (id n) is used at least once: dimension == n1-1 and Q1 is exercised
    if all(i != n1 - 1 for row in rows for i, _ in row):
        rows[0].append((n1 - 1, F(1, 2)))
    return make_case(name, w0, w, V, rows, ys, fr(1, 5, 20), fr(0, 3, 100), fr(0, 3, 100), fr(1, 5, 100),
                     (fr(0, 2, 10), fr(0, 5, 10), fr(1, 20, 2)))


def main():
    cases = [
        survey_case(),
        random_case("rand_k3_unsorted", 11, 3, 7, 6, 5),
        random_case("rand_k1_single", 12, 1, 5, 5, 1),        # single-nnz rows: zero interaction (Q6)
        random_case("rand_k4_emptyrow", 13, 4, 9, 7, 6, empty_row=True),
        random_case("rand_k8_wide", 14, 8, 12, 10, 8, unsorted=False),
    ]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_fm.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/make_kats.py", "cases": cases}, fh, indent=1)
    print("wrote", out, "cases:", [c["name"] for c in cases])


if __name__ == "__main__":
    main()
