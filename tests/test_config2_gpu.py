"""configs[2] at FULL size inside the -m gpu suite: 1 000 000 synthetic ads d = 256, the benchmark's
own models / corpus / ad table / users (bench.build_models, bench.device_corpus), one batch of 512 users through
UserTower -> exact top-500 -> TransformerRanker -> top-10, the first 16 users checked end to end against
oracle.pipeline.recommend with bench.parity_check (the same check every default bench.py run prints): stage-1 sets
tolerance-aware, logits against the STRICT 1e-4 * max(1, |logit|) rule, top-10 selection exact and equal to the oracle's.
Both ranker engines: the default f16x3 row-owner kernel and strict fp32 MFMA."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_config2_one_million_ads_end_to_end_vs_oracle():
    import bench
    from amdrec import synth
    from amdrec.index import FAISSIndex
    from amdrec.pipeline import AdRecommenderInference
    dev = torch.device("cuda", 0)
    tt, rk, (tt_sd, rk_sd), dims = bench.build_models(dev)
    user, ad, nnum = dims
    corpus = bench.device_corpus(bench.N_ADS, bench.DIM, dev)
    index = FAISSIndex(bench.DIM, index_type="Flat", device=dev)
    index.add(corpus)
    del corpus
    ad_table_np = synth.ad_features(ad, bench.N_ADS, seed=99)
    ad_table = torch.from_numpy(ad_table_np).to(dev)
    rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index, ad_features=ad_table)
    uc_np, un_np = synth.user_batch(user, nnum, bench.USERS_PER_GPU, seed=2024)
    uc, un = torch.from_numpy(uc_np).to(dev), torch.from_numpy(un_np).to(dev)
    n_check = 16
    _, ref = bench.cpu_baseline(tt_sd, rk_sd, dims, index._xb[:index._n].cpu().numpy(), ad_table_np, uc_np, un_np, n_check,
                               torch_leg=False)
    for engine in ("f16x3", "fp32"):
        rk.gemm_engine = engine
        assert rk.gemm_engine_for(bench.USERS_PER_GPU * bench.STAGE1_K) == engine
        out = rec.recommend_device(uc, un, bench.TOP_K, bench.STAGE1_K)
        torch.cuda.synchronize()
        assert out["ad_ids"].shape == (bench.USERS_PER_GPU, bench.TOP_K)
        par = bench.parity_check(ref, out, n_check)
        assert par["topk_set_ok"] and par["top10_selection_exact"], (engine, par)
        assert par["max_logit_err_over_bound"] <= 1.0, (engine, par)
        assert par["top10_equal_to_oracle_frac"] == 1.0, (engine, par)
        # every user of the batch: unique candidates, scores sorted, top-10 drawn from the user's own candidates
        cand, cs = out["candidate_ids"], out["candidate_scores"]
        assert bool((cs[:, :-1] >= cs[:, 1:]).all())
        assert all(len(set(r)) == bench.STAGE1_K for r in cand[::37].cpu().numpy().tolist())
        assert bool((out["ad_ids"].unsqueeze(2) == cand.unsqueeze(1)).any(2).all())
