#!/usr/bin/env python3
"""CPU study of error-compensated split GEMMs for the ranker (no GPU needed).

Emulates, in numpy, the arithmetic of three ways of computing an fp32 GEMM on the 16-bit MFMA
(products of two 16-bit values are exact in fp32; accumulation is fp32):
  x6t  : bf16 truncation split h+m+l (8+8+8 bits), products hh hm mh mm hl lh            (round-1 kernel)
  x6r  : bf16 round-to-nearest split (each plane = RN(residual)), same six products
  x3h  : fp16 round-to-nearest split h+l (11+11 bits) of power-of-two pre-scaled operands, products hh hl lh
and runs the whole TransformerRanker trunk + heads with each, comparing logits against float64.
The fp32 accumulation is emulated by float32 matmuls of the plane matrices (numpy sgemm: blocked order,
fp32-level), so the numbers below bound the SPLIT error, which is what is being chosen; the real kernels'
accumulation error is measured on the GPU by tools/x6_probe.hip / tests.

    python tools/split_accuracy.py            # prints a JSON summary
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))

from amdrec import synth  # noqa: E402


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def bf16_rn(x):
    u = x.view(np.uint32)
    r = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return r.view(np.float32)


def split3(x, rnd):
    x = np.ascontiguousarray(x, dtype=np.float32)
    h = rnd(x)
    r1 = x - h
    m = rnd(r1)
    r2 = r1 - m
    l = rnd(r2)
    return h, m, l


def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32).T).astype(np.float32)


def gemm_x6(x, w, rnd):
    xh, xm, xl = split3(x, rnd)
    wh, wm, wl = split3(w, rnd)
    # order of the kernel: small terms first
    acc = mm32(xl, wh) + mm32(xh, wl)
    acc = acc + mm32(xm, wm)
    acc = acc + mm32(xm, wh) + mm32(xh, wm)
    return (acc + mm32(xh, wh)).astype(np.float32)


def pow2_scale(maxabs, target_exp):
    """power of two s with maxabs * s in [2^(target_exp-1), 2^target_exp)"""
    if maxabs == 0:
        return 1.0
    e = np.floor(np.log2(maxabs)) + 1
    return float(2.0 ** (target_exp - e))


def split_f16(x):
    h = x.astype(np.float16).astype(np.float32)
    l = (x - h).astype(np.float16).astype(np.float32)
    return h, l


def gemm_x3(x, w, row_scale=True):
    """fp16 two-plane split; x rows scaled per row (power of two), w per matrix."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    if row_scale:
        mx = np.abs(x).max(axis=1)
        sx = np.array([pow2_scale(v, 8) for v in mx], dtype=np.float32)[:, None]
    else:
        sx = np.float32(pow2_scale(np.abs(x).max(), 8))
    sw = np.float32(pow2_scale(np.abs(w).max(), 4))
    xh, xl = split_f16(x * sx)
    wh, wl = split_f16(w * sw)
    assert np.isfinite(xh).all() and np.isfinite(wh).all()
    acc = mm32(xl, wh) + mm32(xh, wl)
    acc = acc + mm32(xh, wh)
    return (acc / (sx * sw)).astype(np.float32)


def gemm_f32(x, w):
    return mm32(x, w)


def gemm_f64(x, w):
    return x.astype(np.float64) @ w.astype(np.float64).T


def layer_norm(x, g, b, dt):
    x = x.astype(dt)
    mu = x.mean(axis=1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=1, keepdims=True)
    return (xc / np.sqrt(var + dt(1e-5)) * g.astype(dt) + b.astype(dt)).astype(dt)


def ranker(sd, feats, gemm, dt):
    """transformer_ranker.py:332-380 with every Linear going through `gemm` (fused W_ov as the product does)."""
    c = lambda a: np.asarray(a).astype(dt)   # noqa: E731
    x = gemm(feats, sd["feature_projection.weight"]).astype(dt) + c(sd["feature_projection.bias"]) + \
        c(sd["positional_encoding"][0, 0])
    l = 0
    while f"transformer_layers.{l}.norm1.weight" in sd:
        p = f"transformer_layers.{l}"
        wv, bv = sd[p + ".self_attention.W_v.weight"].astype(np.float64), sd[p + ".self_attention.W_v.bias"].astype(np.float64)
        wo, bo = sd[p + ".self_attention.W_o.weight"].astype(np.float64), sd[p + ".self_attention.W_o.bias"].astype(np.float64)
        wov, bov = (wo @ wv).astype(np.float32), (wo @ bv + bo).astype(np.float32)
        a = gemm(x.astype(np.float32) if dt is np.float32 else x, wov).astype(dt) + c(bov)
        x = layer_norm(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], dt)
        h = np.maximum(gemm(x, sd[p + ".feed_forward.fc1.weight"]).astype(dt) + c(sd[p + ".feed_forward.fc1.bias"]), 0)
        f = gemm(h, sd[p + ".feed_forward.fc2.weight"]).astype(dt) + c(sd[p + ".feed_forward.fc2.bias"])
        x = layer_norm(x + f, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], dt)
        l += 1
    x0, xl = x, x
    i = 0
    while f"feature_interaction.cross_weights.{i}" in sd:
        w = sd[f"feature_interaction.cross_weights.{i}"]
        xl = (x0 * (gemm(xl, np.ascontiguousarray(w.T)).astype(dt) + c(sd[f"feature_interaction.cross_biases.{i}"])) + xl).astype(dt)
        i += 1
    out = {}
    for t in ("ctr", "engagement", "revenue"):
        p = f"prediction_heads.{t}"
        h = np.maximum(gemm(xl, sd[p + ".0.weight"]).astype(dt) + c(sd[p + ".0.bias"]), 0)
        h = np.maximum(gemm(h, sd[p + ".3.weight"]).astype(dt) + c(sd[p + ".3.bias"]), 0)
        out[t] = (gemm(h, sd[p + ".6.weight"]).astype(dt) + c(sd[p + ".6.bias"]))[:, 0]
    return out


def feats_of(sd, user, ad, nnum, rows, seed):
    import oracle
    rng = np.random.default_rng(seed)
    uc = np.stack([rng.integers(0, c, rows) for c in user.values()], axis=1)
    ac = np.stack([rng.integers(0, c, rows) for c in ad.values()], axis=1)
    un = rng.standard_normal((rows, nnum)).astype(np.float32)
    return oracle.ranker.embed_features(sd, uc, ac, un)


def main():
    user, ad, nnum = synth.demo_dims()
    res = {}
    # 1. single GEMM, random data (as tools/x6_probe.hip): max |err| vs fp64, K = 256 / 1024 / 4096
    rng = np.random.default_rng(0)
    for K in (256, 1024, 4096):
        x = rng.standard_normal((512, K)).astype(np.float32)
        w = (rng.standard_normal((256, K)) / np.sqrt(K)).astype(np.float32)
        ref = gemm_f64(x, w)
        row = {}
        for name, fn in (("fp32", gemm_f32), ("x6t", lambda a, b: gemm_x6(a, b, bf16_trunc)),
                         ("x6r", lambda a, b: gemm_x6(a, b, bf16_rn)), ("x3h", gemm_x3)):
            e = fn(x, w).astype(np.float64) - ref
            row[name] = {"max": float(np.abs(e).max()), "rms": float(np.sqrt((e * e).mean())),
                         "mean": float(e.mean())}
        res[f"gemm_K{K}"] = row
    # 2. whole ranker, both weight scalings, 2000 rows: logit error vs fp64, relative to the strict bound
    for cross, scale in (("scaled", 1.0 / 16), ("randn", 1.0)):
        sd = synth.ranker_state(user, ad, nnum, seed=12, cross_scale=scale)
        feats = feats_of(sd, user, ad, nnum, 2000, seed=3)
        ref = ranker(sd, feats, gemm_f64, np.float64)
        scale_all = max(np.abs(v).max() for v in ref.values())
        row = {"max_abs_logit": float(scale_all)}
        for name, fn in (("fp32", gemm_f32), ("x6t", lambda a, b: gemm_x6(a, b, bf16_trunc)),
                         ("x6r", lambda a, b: gemm_x6(a, b, bf16_rn)), ("x3h", gemm_x3)):
            got = ranker(sd, feats, fn, np.float32)
            worst, worst_scale = 0.0, 0.0
            for t in ref:
                err = np.abs(got[t].astype(np.float64) - ref[t])
                worst = max(worst, float((err / (1e-4 * np.maximum(1.0, np.abs(ref[t])))).max()))
                worst_scale = max(worst_scale, float(err.max() / scale_all))
            row[name] = {"max_err_over_strict_bound": worst, "max_err_over_batch_scale": worst_scale}
        res[f"ranker_{cross}"] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
