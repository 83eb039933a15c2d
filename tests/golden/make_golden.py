#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE modules in this container.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It imports ``two_tower_model.py`` and ``transformer_ranker.py`` from /root/reference
(they import cleanly: torch only), loads weights drawn by this repo's own seeded
generator (amdrec.synth), runs the eval-mode forward on CPU and writes small ``.npz``
fixtures (inputs, outputs, sha256 of the weights) next to this script.  The reference
sources never travel; only these data files do.  ``faiss_retrieval.py`` / ``inference.py``
cannot be imported (``import faiss`` fails: ModuleNotFoundError), so the search stage has
no fixture from the reference (parity unpinned there, see oracle/search.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
sys.path.insert(0, "/root/reference")

from amdrec import synth  # noqa: E402
import two_tower_model as ref_tt  # noqa: E402
import transformer_ranker as ref_rk  # noqa: E402


def to_torch(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def small_dims():
    """Ragged small cardinalities incl. the C25/C26 = 20/10 tail of the synthetic data."""
    cards = synth.CRITEO_SYNTH_CARDS
    from collections import OrderedDict
    user = OrderedDict((c, min(cards[i], 174 + 7 * i)) for i, c in enumerate(synth.USER_COLS))
    ad = OrderedDict((c, min(cards[6 + i], 97 + 13 * i)) for i, c in enumerate(synth.AD_COLS))
    return user, ad, 13


CASES = {
    # name: (dims fn, batch sizes, seed)
    "demo": (synth.demo_dims, (1, 7, 64), 11),
    "ragged": (small_dims, (3, 33), 12),
    # the reference's SECOND usage example (tutorial.ipynb cells 10 and 19): smaller constructor arguments than the
    # defaults, on preprocessor-sized (ragged) feature dims
    "tutorial": (small_dims, (1, 7, 64), 13),
}
# constructor arguments that differ from the reference's defaults (two_tower_model.py:193-201, transformer_ranker.py:213-224)
ARCH = {
    "tutorial": {"tt": dict(embedding_dim=16, hidden_dims=[256, 128], output_dim=128),
                 "rk": dict(embedding_dim=16, d_model=128, num_heads=4, num_layers=2, d_ff=512)},
}


def main(only=None):
    torch.set_num_threads(1)
    torch.manual_seed(0)
    for name, (dims_fn, batches, seed) in CASES.items():
        if only and name not in only:
            continue
        user_dims, ad_dims, nnum = dims_fn()
        arch = ARCH.get(name, {"tt": {}, "rk": {}})
        # ---- two-tower -------------------------------------------------------------
        sd = synth.two_tower_state(user_dims, ad_dims, nnum, seed=seed, **arch["tt"])
        model = ref_tt.TwoTowerModel(dict(user_dims), dict(ad_dims), nnum, **arch["tt"])
        model.load_state_dict(to_torch(sd))
        model.eval()
        out = {"weights_sha256": synth.state_sha256(sd), "seed": seed}
        for B in batches:
            ucat, unum = synth.user_batch(user_dims, nnum, B, seed=seed + B)
            acat = synth.ad_features(ad_dims, B, seed=seed + 100 + B)
            with torch.no_grad():
                ue, ae = model(torch.from_numpy(ucat), torch.from_numpy(unum),
                               torch.from_numpy(acat))
                ps = model.predict_scores(torch.from_numpy(ucat), torch.from_numpy(unum),
                                          torch.from_numpy(acat))
            out[f"B{B}_user_cat"] = ucat.astype(np.int16)
            out[f"B{B}_user_num"] = unum
            out[f"B{B}_ad_cat"] = acat.astype(np.int16)
            out[f"B{B}_user_emb"] = ue.numpy()
            out[f"B{B}_ad_emb"] = ae.numpy()
            out[f"B{B}_scores"] = ps.numpy()
        np.savez_compressed(os.path.join(HERE, f"two_tower_{name}.npz"), **out)

        # ---- ranker ----------------------------------------------------------------
        for cs_name, cross_scale in (("randn", 1.0), ("scaled", 1.0 / 16)):
            sd = synth.ranker_state(user_dims, ad_dims, nnum, seed=seed + 1,
                                    cross_scale=cross_scale, **arch["rk"])
            model = ref_rk.TransformerRanker(dict(user_dims), dict(ad_dims), nnum, **arch["rk"])
            model.load_state_dict(to_torch(sd))
            model.eval()
            out = {"weights_sha256": synth.state_sha256(sd), "seed": seed + 1,
                   "cross_scale": cross_scale}
            for B in batches:
                ucat, unum = synth.user_batch(user_dims, nnum, B, seed=seed + 200 + B)
                acat = synth.ad_features(ad_dims, B, seed=seed + 300 + B)
                with torch.no_grad():
                    pred = model(torch.from_numpy(ucat), torch.from_numpy(acat),
                                 torch.from_numpy(unum))
                out[f"B{B}_user_cat"] = ucat.astype(np.int16)
                out[f"B{B}_user_num"] = unum
                out[f"B{B}_ad_cat"] = acat.astype(np.int16)
                assert list(pred.keys()) == ["ctr", "engagement", "revenue"]
                for t, v in pred.items():
                    out[f"B{B}_{t}"] = v.numpy()
            np.savez_compressed(os.path.join(HERE, f"ranker_{name}_{cs_name}.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main(sys.argv[1:])            # optional: case names to (re)generate; default all
