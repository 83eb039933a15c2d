"""The fp16x3 row-owner engine (csrc/rowowner.hpp) phase by phase: amdrec_ranker_x3_prefix runs the first n phases of
the chain on projected rows and returns the row state, which is compared with a float64 evaluation of the same prefix
(oracle.ranker.chain_states).  An error is thereby localised to one phase (attention + LN1, FFN + LN2, a cross layer,
the heads) instead of showing up as a wrong logit."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu


VARIANT = 32


@pytest.fixture(params=[32, 16], autouse=True)
def variant(request):
    """Every case runs on both row-owner kernels: 32 rows per wave (rowowner.hpp) and 16 rows per wave (rowowner16.hpp)."""
    global VARIANT
    VARIANT = request.param
    yield request.param
    VARIANT = 32


def _model(name, cross):
    from amdrec.ranker import TransformerRanker
    user, ad, nnum, sd, _ = cases.ranker_case(name, cross)
    m = TransformerRanker(dict(user), dict(ad), nnum)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.x3_variant = VARIANT
    return m.cuda().eval(), sd, (user, ad, nnum)


def _projected_rows(sd, dims, rows, seed):
    user, ad, nnum = dims
    uc, un = synth.user_batch(user, nnum, rows, seed=seed)
    ac = synth.ad_features(ad, rows, seed=seed + 1)
    feats = oracle.ranker.embed_features(sd, uc, ac, un)
    return (feats @ sd["feature_projection.weight"].T + sd["feature_projection.bias"]
            + sd["positional_encoding"][0, 0]).astype(np.float32)


def _prefix(m, X, n_phases):
    from amdrec import _lib
    lib = _lib.load()
    dev = X.device
    params, tasks = m._pack(dev)
    rows = X.shape[0]
    x_out = torch.full((rows, 256), float("nan"), dtype=torch.float32, device=dev)
    logits = torch.full((len(tasks), rows), float("nan"), dtype=torch.float32, device=dev)
    ws = torch.empty(((rows + 127) // 128) * 128 * 1024, dtype=torch.uint8, device=dev)
    _lib.check(lib.amdrec_ranker_x3_prefix(C.byref(params), _lib.ptr(X), X.stride(0), rows, n_phases, _lib.ptr(x_out),
                                           x_out.stride(0), _lib.ptr(logits), logits.stride(0), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    return x_out.cpu().numpy(), logits.cpu().numpy()


@pytest.mark.parametrize("rows", [128 * 3 + 45,            # <= 4096 rows: the column-split kernel (x3c, variant 16), ragged tail
                                  4096 + 64 * 5 + 13,      # <= 16384 rows: the 64-row workgroup shape (x3b4), ragged tail
                                  16384 + 128 * 3 + 45])   # beyond: the 128-row shape (x3b), three full workgroups + a ragged one
@pytest.mark.parametrize("cross", ["scaled", "randn"])
def test_every_prefix_of_the_chain_matches_float64(cross, rows, accuracy):
    m, sd, dims = _model("demo", cross)
    assert m.gemm_engine == "f16x3" and m.gemm_engine_for(10_000) == "f16x3"
    X = _projected_rows(sd, dims, rows, seed=41)
    truth = oracle.ranker.chain_states(sd, X, dtype=np.float64)
    f32 = oracle.ranker.chain_states(sd, X, dtype=np.float32)
    Xd = torch.from_numpy(X).cuda()
    n_total = len(truth)
    names = [f"L{l}.{k}" for l in range(3) for k in ("attn_ln1", "ffn_ln2")] + [f"cross{c}" for c in range(3)] + ["heads"]
    assert n_total == len(names) == 10
    for n in range(1, n_total + 1):
        x, logits = _prefix(m, Xd, n)
        if n < n_total:
            ref = truth[n - 1]
            scale = np.abs(ref).max(axis=1, keepdims=True)            # per-row magnitude
            d, d32 = (x - ref) / scale, (f32[n - 1] - ref) / scale    # the engine's / the numpy fp32 evaluation's error
            err, err32 = float(np.abs(d).max()), float(np.abs(d32).max())
            rms, rms32 = float(np.sqrt(np.mean(d * d))), float(np.sqrt(np.mean(d32 * d32)))
            accuracy(f"x3_prefix/demo_{cross}/rows{rows}/{names[n - 1]}", f"f16x3/{VARIANT}", err / max(err32, 1e-30),
                     rel_err_vs_float64=err, numpy_fp32_rel_err_vs_float64=err32, rms_ratio=rms / max(rms32, 1e-30),
                     rms_rel_err_vs_float64=rms, numpy_fp32_rms_rel_err_vs_float64=rms32)
            assert np.isfinite(x).all(), names[n - 1]
            # Mid-chain error of the emulation against the numpy fp32 evaluation of the same prefix, both measured against
            # float64.  What the arithmetic allows: a MAC of the engine carries three terms of up to 2^-22 (the two operands'
            # split errors and the dropped l*l product) = 12 * 2^-24 where an fp32 fma chain carries two roundings = 2 * 2^-24,
            # i.e. a ratio of the BOUNDS of 6.  The RMS over the ~110k elements of a phase is the stable statistic and is
            # held to 4x; the ratio of the two MAXIMA is noisy (the fp32 evaluation's realised maximum moves by +-30 % with
            # the seed and sits well under its own bound) and is held to 8x, not to the 6x that round 3 had fitted to a
            # recorded worst of 5.61 ("a bound fitted to the observation").
            assert rms <= 4 * rms32, (names[n - 1], rms, rms32)
            assert err <= 8 * err32, (names[n - 1], err, err32)
        else:
            scale = cases.logit_scale(truth[-1])
            for ti, t in enumerate(oracle.ranker.TASKS):
                ok, e = cases.logit_close(logits[ti], truth[-1][t], cross, scale=scale)
                accuracy(f"x3_prefix/demo_{cross}/rows{rows}/heads/{t}", f"f16x3/{VARIANT}", e)
                assert ok, (t, e)


def test_x3_whole_forward_small_batches_match_reference_golden(accuracy):
    """x3_min_rows = 1 drives the reference's golden batches (1, 7, 64 rows) through the row-owner kernel."""
    from tests.conftest import load_golden
    for cross in ("scaled", "randn"):
        m, sd, _ = _model("demo", cross)
        m.x3_min_rows = 1
        g = load_golden(f"ranker_demo_{cross}.npz")
        for B in (1, 7, 64):
            assert m.gemm_engine_for(B) == "f16x3"
            cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
            pred = m(cu(g[f"B{B}_user_cat"]), cu(g[f"B{B}_ad_cat"]), cu(g[f"B{B}_user_num"]))
            scale = cases.logit_scale({t: g[f"B{B}_{t}"] for t in pred})
            for t in pred:
                ok, err = cases.logit_close(pred[t].cpu().numpy(), g[f"B{B}_{t}"], cross, scale=scale)
                accuracy(f"golden/demo_{cross}/B{B}/{t}", f"f16x3/{VARIANT}", err)
                assert ok, (cross, B, t, err)


def test_x3_extreme_rows_do_not_overflow():
    """Rows spanning 60 binades (all-zero, 1e-30, 1e+6 scales) stay finite: the per-row power-of-two scaling keeps every
    fp16 plane in range whatever the row's magnitude."""
    m, sd, dims = _model("demo", "scaled")
    X = _projected_rows(sd, dims, 256, seed=43)
    X[0] = 0.0
    X[1] *= 1e-30
    X[2] *= 1e6
    X[3, :] = 0.0
    X[3, 17] = 5e4
    truth = oracle.ranker.chain_states(sd, X, dtype=np.float64)
    x, _ = _prefix(m, torch.from_numpy(X).cuda(), 2)
    assert np.isfinite(x).all()
    ref = truth[1]
    assert (np.abs(x - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= 2e-5


def test_x3_degenerate_weights_zero_ffn_and_zero_cross():
    """A layer whose fc1 is all zero (hidden bound 0: the hidden-tile scale sits on its clamp) and a cross layer with zero
    weights: the scaled domains must stay finite and the logits match float64."""
    from amdrec.ranker import TransformerRanker
    user, ad, nnum, sd, _ = cases.ranker_case("demo", "scaled")
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd["transformer_layers.0.feed_forward.fc1.weight"][:] = 0
    sd["transformer_layers.0.feed_forward.fc1.bias"][:] = 0
    sd["feature_interaction.cross_weights.1"][:] = 0
    m = TransformerRanker(dict(user), dict(ad), nnum)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.x3_variant = VARIANT
    m = m.cuda().eval()
    X = _projected_rows(sd, (user, ad, nnum), 300, seed=47)
    truth = oracle.ranker.chain_states(sd, X, dtype=np.float64)
    n_phases = len(truth)
    x, logits = _prefix(m, torch.from_numpy(X).cuda(), n_phases)
    assert np.isfinite(logits).all()
    scale = cases.logit_scale(truth[-1])
    for ti, t in enumerate(oracle.ranker.TASKS):
        ok, err = cases.logit_close(logits[ti], truth[-1][t], "scaled", scale=scale)
        assert ok, (t, err)
    # and the state right after the degenerate FFN (phase 2) and the zero cross layer (phase 8) matches float64 row-wise
    for n in (2, 8):
        x, _ = _prefix(m, torch.from_numpy(X).cuda(), n)
        ref = truth[n - 1]
        assert np.isfinite(x).all() and (np.abs(x - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= 2e-5, n


@pytest.mark.parametrize("rows", [500, 16 * 3 + 5, 4096, 1])
def test_column_split_kernel_is_bit_identical_to_the_16_row_kernel(rows):
    """csrc/rowowner16c.hpp (four waves split the output features of 16 rows) against csrc/rowowner16.hpp (a wave owns its
    16 rows alone): the same MFMAs in the same order per output element, the same row-wise functions on the same values -
    every prefix of the chain and the logits must be EQUAL, not close."""
    if VARIANT != 16:
        pytest.skip("the column-split kernel belongs to the 16-row variant")
    from amdrec import _lib
    m, sd, dims = _model("demo", "randn")
    X = torch.from_numpy(_projected_rows(sd, dims, rows, seed=47)).cuda()
    lib = _lib.load()
    for n in list(range(1, 11)):
        m.x3_cs_max_rows = -1
        x_ref, l_ref = _prefix(m, X, n)
        m.x3_cs_max_rows = 0
        _lib.check(lib.amdrec_profile_enable(1))
        x_cs, l_cs = _prefix(m, X, n)
        tags = list(_lib.profile_report())
        _lib.check(lib.amdrec_profile_enable(0))
        assert tags == ["ranker_colsplit16_x3"], tags
        if n < 10:
            assert np.array_equal(x_ref, x_cs), (n, np.abs(x_ref - x_cs).max())
        else:
            assert np.isfinite(l_ref).all() and np.array_equal(l_ref, l_cs), np.abs(l_ref - l_cs).max()
