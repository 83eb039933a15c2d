"""Dev tool: where the HOST time of one reference-API request goes (cProfile over 300 recommend_ads calls on configs[0]'s
model_dir).  usage: python tools/api_host_profile.py"""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
from amdrec import prep, synth  # noqa: E402
from amdrec.pipeline import AD_COLS, USER_COLS, AdRecommenderInference, build_faiss_index  # noqa: E402
from amdrec.towers import TwoTowerModel  # noqa: E402

t = lambda sd: {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}   # noqa: E731
numerical, categorical, labels = prep.synthetic_criteo(10000)
pp, num_scaled, cat_enc = prep.fit_preprocessor(numerical, categorical)
user_dims = {c: pp.feature_dims[c] for c in USER_COLS}
ad_dims = {c: pp.feature_dims[c] for c in AD_COLS}
ad_table = cat_enc[:7000, 6:]
tt_sd = synth.two_tower_state(user_dims, ad_dims, 13, seed=51)
rk_sd = synth.ranker_state(user_dims, ad_dims, 13, seed=52, cross_scale=1.0 / 16)
with tempfile.TemporaryDirectory() as d:
    pp.save(os.path.join(d, "preprocessor.json"))
    torch.save({"model_state_dict": t(tt_sd)}, os.path.join(d, "two_tower_best.pt"))
    torch.save(t(rk_sd), os.path.join(d, "transformer_ranker_final.pt"))
    np.save(os.path.join(d, "ad_features.npy"), ad_table)
    tt = TwoTowerModel(user_dims, ad_dims, 13)
    tt.load_state_dict(t(tt_sd))
    build_faiss_index(tt, ad_table, save_path=os.path.join(d, "faiss_index.bin"))
    rec = AdRecommenderInference(d)
rng = np.random.default_rng(7)
users = [{"categorical": {f"C{i}": f"cat_{rng.integers(0, 50)}" for i in range(1, 7)},
          "numerical": {f"I{i}": float(rng.random() * 100) for i in range(1, 14)}} for _ in range(300)]
for u in users[:10]:
    rec.recommend_ads(u)
pr = cProfile.Profile()
pr.enable()
for u in users:
    rec.recommend_ads(u)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
