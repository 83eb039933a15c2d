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
}
CROSS = {"randn": 1.0, "scaled": 1.0 / 16}

# Parity tolerances, stated once and used by every parity test:
EMB_ATOL = 1e-5          # unit-norm tower embeddings, absolute
LOGIT_RTOL = 1e-4        # ranker logits: |d| <= LOGIT_RTOL * max(1, |logit|)
SCORE_ATOL = 1e-6        # inner-product scores of unit vectors, absolute
TOPK_TAU = 1e-5          # near-tie band at the k-th score


def two_tower_case(name):
    dims_fn, batches, seed = CASES[name]
    user, ad, nnum = dims_fn()
    sd = synth.two_tower_state(user, ad, nnum, seed=seed)
    return user, ad, nnum, sd, batches


def ranker_case(name, cross):
    dims_fn, batches, seed = CASES[name]
    user, ad, nnum = dims_fn()
    sd = synth.ranker_state(user, ad, nnum, seed=seed + 1, cross_scale=CROSS[cross])
    return user, ad, nnum, sd, batches


LOGIT_SCALE_RTOL = 1e-5  # ... or |d| <= LOGIT_SCALE_RTOL * max|logit| over the batch


def logit_scale(ref_dict):
    """Largest |logit| over the batch and over all tasks (the heads share one trunk, whose
    magnitude sets the fp32 rounding noise of every head)."""
    import numpy as np
    return max(float(np.abs(np.asarray(v)).max()) if np.asarray(v).size else 0.0 for v in ref_dict.values())


def logit_close(got, ref, rtol=LOGIT_RTOL, scale_rtol=LOGIT_SCALE_RTOL, scale=None):
    """Ranker-logit tolerance.  Two fp32 evaluations of the reference net with its default
    unscaled randn cross weights (transformer_ranker.py:177-180) differ by up to 4e-4
    relative per element (the reference's own torch output vs float64 truth, measured in
    the build container: heads cancel terms of magnitude ~1e3) while staying within 2.5e-6
    of the batch's logit scale; so an element passes if it is within ``rtol`` of itself
    OR within ``scale_rtol`` of the largest |logit| of the batch (over all tasks: ``scale``)."""
    import numpy as np
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if scale is None:
        scale = float(np.abs(ref).max()) if ref.size else 0.0
    bound = np.maximum(rtol * np.maximum(1.0, np.abs(ref)), scale_rtol * scale)
    err = np.abs(got - ref)
    return bool((err <= bound).all()), float((err / bound).max()) if ref.size else 0.0
