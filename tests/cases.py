"""Shared case definitions for the golden fixtures (mirrors tests/golden/make_golden.py)."""
from collections import OrderedDict

from amdrec import synth


def small_dims():
    cards = synth.CRITEO_SYNTH_CARDS
    user = OrderedDict((c, min(cards[i], 174 + 7 * i)) for i, c in enumerate(synth.USER_COLS))
    ad = OrderedDict((c, min(cards[6 + i], 97 + 13 * i)) for i, c in enumerate(synth.AD_COLS))
    return user, ad, 13


CASES = {
    "demo": (synth.demo_dims, (1, 7, 64), 11),
    "ragged": (small_dims, (3, 33), 12),
    # the reference's second usage example (tutorial.ipynb cells 10, 19): NOT the default architecture
    "tutorial": (small_dims, (1, 7, 64), 13),
}
# constructor arguments that differ from the reference's defaults (two_tower_model.py:193-201, transformer_ranker.py:213-224)
ARCH = {
    "tutorial": {"tt": dict(embedding_dim=16, hidden_dims=[256, 128], output_dim=128),
                 "rk": dict(embedding_dim=16, d_model=128, num_heads=4, num_layers=2, d_ff=512)},
}


def arch(name):
    return ARCH.get(name, {"tt": {}, "rk": {}})
CROSS = {"randn": 1.0, "scaled": 1.0 / 16}

# Parity tolerances, stated once and used by every parity test:
EMB_ATOL = 1e-5          # unit-norm tower embeddings, absolute
LOGIT_RTOL = 1e-4        # ranker logits: |d| <= LOGIT_RTOL * max(1, |logit|)
SCORE_ATOL = 1e-6        # inner-product scores of unit vectors, absolute
TOPK_TAU = 1e-5          # near-tie band at the k-th score


def two_tower_case(name):
    dims_fn, batches, seed = CASES[name]
    user, ad, nnum = dims_fn()
    sd = synth.two_tower_state(user, ad, nnum, seed=seed, **arch(name)["tt"])
    return user, ad, nnum, sd, batches


def ranker_case(name, cross):
    dims_fn, batches, seed = CASES[name]
    user, ad, nnum = dims_fn()
    sd = synth.ranker_state(user, ad, nnum, seed=seed + 1, cross_scale=CROSS[cross], **arch(name)["rk"])
    return user, ad, nnum, sd, batches


LOGIT_SCALE_RTOL = 1e-5  # randn fixtures only: ... or |d| <= LOGIT_SCALE_RTOL * max|logit| over the batch
LOGIT_STRICT_RTOL = 1e-5  # scaled fixtures (the benchmark's weights): |d| <= 1e-5 * max(1, |logit|), 10x inside SURVEY 8a


def logit_scale(ref_dict):
    """Largest |logit| over the batch and over all tasks (the heads share one trunk, whose
    magnitude sets the fp32 rounding noise of every head)."""
    import numpy as np
    return max(float(np.abs(np.asarray(v)).max()) if np.asarray(v).size else 0.0 for v in ref_dict.values())


def logit_close(got, ref, cross="randn", scale=None):
    """Ranker-logit tolerance, by conditioning of the network:

    * ``cross == "scaled"`` (cross weights randn/16: logits O(1), the weights bench.py uses): the STRICT rule alone,
      |d| <= LOGIT_STRICT_RTOL * max(1, |logit|) = 1e-5 - ten times tighter than SURVEY 8a's 1e-4; two fp32
      evaluations of this network sit ~1e-6 apart (measured: oracle vs reference golden 0.6-0.9e-6).
    * ``cross == "randn"`` (the reference's default unscaled randn(256,256) cross weights,
      transformer_ranker.py:177-184: logits ~1e3 out of three 16x-amplifying layers, heads cancel): the reference's
      OWN torch fp32 output is up to 4.4e-4 relative away from float64 truth per element while staying within 2.5e-6
      of the batch's logit scale, so an element passes if it is within SURVEY 8a's 1e-4 * max(1,|logit|) OR within
      LOGIT_SCALE_RTOL of the largest |logit| of the batch (``scale``, over all tasks).
    -> (ok, max err/bound)"""
    import numpy as np
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if cross == "scaled":
        bound = LOGIT_STRICT_RTOL * np.maximum(1.0, np.abs(ref))
    else:
        if scale is None:
            scale = float(np.abs(ref).max()) if ref.size else 0.0
        bound = np.maximum(LOGIT_RTOL * np.maximum(1.0, np.abs(ref)), LOGIT_SCALE_RTOL * scale)
    err = np.abs(got - ref)
    return bool((err <= bound).all()), float((err / bound).max()) if ref.size else 0.0
