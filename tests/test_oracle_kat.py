"""The CPU oracle (C restatement + numpy twin) against the exact-rational known-answer
vectors of tests/golden/kat_fm.json (generator: tests/golden/make_kats.py).

The reference has no tests of its own: these KATs, derived from the
naive pairwise FM definition, are what pins the oracle ("parity unpinned by the
reference")."""
import math

import numpy as np
import pytest

import oracle
from oracle import fm_oracle_np as onp
from helpers import f, kat_arrays, random_problem

TOL = dict(rtol=1e-12, atol=1e-13)


def test_kat_file_has_survey_vector(kats):
    c = kats[0]
    assert c["name"] == "survey_s4"
    assert [x["q"] for x in c["yhat"]] == ["-1/50", "24/25", "4/5"]
    assert c["sse"]["q"] == "3161/500"
    assert c["als"]["w0"]["f"] == pytest.approx(44 / 75, rel=1e-15)


def test_predict_residual_rmse(kats):
    for c in kats:
        a = kat_arrays(c)
        yh = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
        np.testing.assert_allclose(yh, f(c["yhat"]), **TOL)
        e = oracle.residual(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"], a["y"])
        np.testing.assert_allclose(e, f(c["e"]), **TOL)
        r = oracle.rmse(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"], a["y"])
        assert r == pytest.approx(math.sqrt(f(c["mse"])), rel=1e-13)
        # numpy twin
        np.testing.assert_allclose(onp.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"]),
                                   f(c["yhat"]), **TOL)


def test_term_q_and_transpose(kats):
    for c in kats:
        a = kat_arrays(c)
        n_rows = len(a["y"])
        cp, rows, cv = oracle.transpose(a["n1"], a["row_ptr"], a["col"], a["val"])
        assert cp[-1] == len(a["col"])
        # inside a column rows ascend; every (row, col, val) triple is preserved
        trip = sorted((int(r), int(i), float(x)) for i in range(a["n1"])
                      for r, x in zip(rows[cp[i]:cp[i + 1]], cv[cp[i]:cp[i + 1]]))
        src = sorted((r, int(a["col"][p]), float(a["val"][p])) for r in range(n_rows)
                     for p in range(a["row_ptr"][r], a["row_ptr"][r + 1]))
        assert trip == src
        for i in range(a["n1"]):
            assert list(rows[cp[i]:cp[i + 1]]) == sorted(rows[cp[i]:cp[i + 1]])
        q_exp = np.array(f(c["q"]))                 # (rows, k)
        for ff in range(a["k"]):
            q = oracle.term_q(a["v"], ff, n_rows, cp, rows, cv)
            np.testing.assert_allclose(q, q_exp[:, ff], **TOL)
        assert oracle.dimension(a["row_ptr"], a["col"]) == a["n1"] - 1


def test_gradient(kats):
    for c in kats:
        a = kat_arrays(c)
        n_rows = len(a["y"])
        for threads in (1, 2):
            gv, gw, g0, sse, e = oracle.batch_grad(a["w0"], a["w"], a["v"], 0, n_rows, a["row_ptr"], a["col"],
                                                   a["val"], a["y"], threads=threads)
            np.testing.assert_allclose(gv, np.array(f(c["grad"]["gV"])), **TOL)
            np.testing.assert_allclose(gw, f(c["grad"]["gw"]), **TOL)
            assert g0 == pytest.approx(f(c["grad"]["g0"]), rel=1e-12, abs=1e-13)
            assert sse == pytest.approx(f(c["sse"]), rel=1e-12)
        gv2, gw2, g02, sse2, _ = onp.batch_grad(a["w0"], a["w"], a["v"], 0, n_rows, a["row_ptr"], a["col"],
                                                a["val"], a["y"])
        np.testing.assert_allclose(gv2, np.array(f(c["grad"]["gV"])), **TOL)
        np.testing.assert_allclose(gw2, f(c["grad"]["gw"]), **TOL)


def test_per_row_h_matches_kat(kats):
    """h_r(v_fi) = x*q_rf - x^2*v_fi (S/fm/lib/ALS.scala:56-58) against d yhat/d v of the naive form."""
    for c in kats:
        a = kat_arrays(c)
        q = np.array(f(c["q"]))
        for r, d in enumerate(c["dyhat_dv"]):
            for p in range(a["row_ptr"][r], a["row_ptr"][r + 1]):
                i, x = int(a["col"][p]), a["val"][p]
                h = x * q[r] - x * x * a["v"][:, i]
                np.testing.assert_allclose(h, f(d[str(i)]), **TOL)


def test_sgd_step(kats):
    for c in kats:
        a = kat_arrays(c)
        s = c["sgd"]
        n_rows = len(a["y"])
        w0, w, v, sse = oracle.sgd_step(a["w0"], a["w"], a["v"], 0, n_rows, a["row_ptr"], a["col"], a["val"],
                                        a["y"], f(s["eta"]), f(s["reg0"]), f(s["regw"]), f(s["regv"]))
        assert w0 == pytest.approx(f(s["w0"]), rel=1e-12, abs=1e-13)
        np.testing.assert_allclose(w, f(s["w"]), **TOL)
        np.testing.assert_allclose(v, np.array(f(s["V"])), **TOL)
        # one whole-dataset batch == one epoch with batch_rows = n_rows
        w0e, we, ve, _ = oracle.sgd_epoch(a["w0"], a["w"], a["v"], n_rows, a["row_ptr"], a["col"], a["val"], a["y"],
                                          f(s["eta"]), f(s["reg0"]), f(s["regw"]), f(s["regv"]))
        np.testing.assert_array_equal(ve, v)
        assert w0e == w0


def test_als_epoch(kats):
    for c in kats:
        a = kat_arrays(c)
        s = c["als"]
        w0, w, v = oracle.als_epoch(a["w0"], a["w"], a["v"], f(s["reg0"]), f(s["regw"]), f(s["regv"]),
                                    a["row_ptr"], a["col"], a["val"], a["y"])
        assert w0 == pytest.approx(f(s["w0"]), rel=1e-10, abs=1e-12)
        np.testing.assert_allclose(w, f(s["w"]), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(v, np.array(f(s["V"])), rtol=1e-9, atol=1e-11)
        # quirk Q1: the last slot (id = num_attribute) is never trained
        assert w[-1] == a["w"][-1]
        np.testing.assert_array_equal(v[:, -1], a["v"][:, -1])


def test_c_oracle_vs_numpy_twin_random():
    a = random_problem(7, 300, 50, 6, 0, 12, empty_rows=(3, 299))
    yh = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
    np.testing.assert_allclose(yh, onp.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"]),
                               rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(yh, onp.predict_pairwise(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"]),
                               rtol=1e-11, atol=1e-12)
    assert yh[3] == a["w0"] and yh[299] == a["w0"]          # empty rows -> w0 (quirk Q6)
    for threads in (1, 3):
        gv, gw, g0, sse, e = oracle.batch_grad(a["w0"], a["w"], a["v"], 10, 250, a["row_ptr"], a["col"], a["val"],
                                               a["y"], threads=threads)
        gv2, gw2, g02, sse2, e2 = onp.batch_grad(a["w0"], a["w"], a["v"], 10, 250, a["row_ptr"], a["col"], a["val"],
                                                 a["y"])
        np.testing.assert_allclose(gv, gv2, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gw, gw2, rtol=1e-10, atol=1e-12)
        assert g0 == pytest.approx(g02, rel=1e-10)
        assert sse == pytest.approx(sse2, rel=1e-10)


def test_training_forward_is_bit_identical_to_predict():
    """The vectorisable loop order of the training path (nonzeros outer) gives the same bits as
    the reference-order predict (factors outer): e from batch_grad == residual."""
    a = random_problem(77, 400, 120, 32, 0, 40, empty_rows=(9,))
    e1 = oracle.residual(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"], a["y"])
    _, _, _, _, e2 = oracle.batch_grad(a["w0"], a["w"], a["v"], 0, 400, a["row_ptr"], a["col"], a["val"], a["y"])
    np.testing.assert_array_equal(e1, e2)


def test_single_nnz_rows_have_zero_interaction():
    a = random_problem(9, 40, 20, 5, 1, 1)
    yh = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
    lin = a["w0"] + a["w"][a["col"]] * a["val"]
    np.testing.assert_array_equal(yh, lin)                  # (vx)^2 - (vx)^2 == 0 exactly (quirk Q6)


def test_finite_difference_gradient():
    a = random_problem(21, 20, 15, 3, 2, 6)
    n_rows = 20

    def loss(v):
        e = oracle.residual(a["w0"], a["w"], v, a["row_ptr"], a["col"], a["val"], a["y"])
        return 0.5 * float((e * e).sum())

    gv, gw, g0, sse, _ = oracle.batch_grad(a["w0"], a["w"], a["v"], 0, n_rows, a["row_ptr"], a["col"], a["val"], a["y"])
    rng = np.random.default_rng(0)
    for _ in range(10):
        ff, i = int(rng.integers(3)), int(rng.integers(15))
        h = 1e-6
        vp, vm = a["v"].copy(), a["v"].copy()
        vp[ff, i] += h
        vm[ff, i] -= h
        fd = (loss(vp) - loss(vm)) / (2 * h)
        assert fd == pytest.approx(gv[ff, i], rel=1e-5, abs=1e-7)


def test_sgd_epoch_order_and_ragged_batches():
    a = random_problem(33, 103, 30, 4, 1, 8)
    args = (a["row_ptr"], a["col"], a["val"], a["y"], 0.05, 0.01, 0.02, 0.03)
    w0, w, v, sse = oracle.sgd_epoch(a["w0"], a["w"], a["v"], 25, *args)
    # same thing as 5 explicit steps (last batch ragged: rows 100..103)
    s0, sw, sv, tot = a["w0"], a["w"], a["v"], 0.0
    for b in range(5):
        s0, sw, sv, s2 = oracle.sgd_step(s0, sw, sv, b * 25, min(103, (b + 1) * 25), *args)
        tot += s2
    np.testing.assert_array_equal(v, sv)
    np.testing.assert_array_equal(w, sw)
    assert w0 == s0 and sse == pytest.approx(tot, rel=1e-15)
    # permuted batch order gives a different trajectory, reproducibly
    o = np.array([4, 2, 0, 1, 3])
    r1 = oracle.sgd_epoch(a["w0"], a["w"], a["v"], 25, *args, order=o)
    r2 = oracle.sgd_epoch(a["w0"], a["w"], a["v"], 25, *args, order=o)
    np.testing.assert_array_equal(r1[2], r2[2])
    assert not np.array_equal(r1[2], v)
    # multi-threaded accumulation agrees to fp64 reassociation
    rt = oracle.sgd_epoch(a["w0"], a["w"], a["v"], 25, *args, threads=4)
    np.testing.assert_allclose(rt[2], v, rtol=1e-12, atol=1e-14)


def test_als_epoch_never_raises_the_regularised_objective():
    """A pin from first principles, independent of how the closed form was restated: the prediction is LINEAR in any single
    parameter with the others held fixed (h = d yhat / d theta does not depend on theta: S/fm/lib/ALS.scala:56-58 gives
    h(v_fi) = x q_f - x^2 v_fi = x (q_f - x v_fi)), so theta* = -(sum e h - theta sum h^2) / (lambda + sum h^2) (:167-176) is the exact
    minimiser of  sum_r e_r^2 + lambda theta^2  along that coordinate — every step, and therefore every epoch, must leave
    J = sum e^2 + reg0 w0^2 + regw |w|^2 + regv |V|^2  (over the trained slots: quirk Q1 leaves the last one alone) no larger."""
    for seed, regs in ((3, (0.0, 0.0, 10.0)), (4, (0.5, 0.1, 2.0)), (5, (0.0, 0.0, 0.0))):
        a = random_problem(seed, 400, 30, 4, 1, 9)
        w0, w, v = a["w0"], a["w"], a["v"]

        def objective(w0, w, v):
            e = oracle.predict(w0, w, v, a["row_ptr"], a["col"], a["val"]) - a["y"]
            return float((e * e).sum() + regs[0] * w0 * w0 + regs[1] * (w[:-1] ** 2).sum() + regs[2] * (v[:, :-1] ** 2).sum())

        j = objective(w0, w, v)
        for _ in range(4):
            w0, w, v = oracle.als_epoch(w0, w, v, regs[0], regs[1], regs[2], a["row_ptr"], a["col"], a["val"], a["y"])
            jn = objective(w0, w, v)
            assert jn <= j * (1 + 1e-12) + 1e-12, (seed, j, jn)
            j = jn
        assert j < objective(a["w0"], a["w"], a["v"])          # and it does learn
