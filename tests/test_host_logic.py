"""CPU tests of the product's HOST-side logic (no kernels): weight packing (BatchNorm fold, K padding,
table stacking, positional-row fold, W_o.W_v pre-multiplication, cross transpose), the preprocessor and the
synthetic-data generator.  The packed weights are evaluated with plain numpy and compared with the oracle."""
import ctypes
import json

import numpy as np
import pytest
import torch

import oracle
from amdrec import prep, synth, weights
from amdrec.pipeline import Preprocessor
from tests import cases


def _arr(pk, ptr, shape, dtype=np.float32):
    """Find the packed (CPU) tensor behind a struct pointer."""
    for t in pk._keep:
        if t.data_ptr() == ptr:
            return t.numpy().reshape(shape)
    raise KeyError(ptr)


def test_tower_packing_folds_batchnorm_exactly():
    user, ad, nnum, sd, _ = cases.two_tower_case("ragged")
    p, pk = weights.pack_tower(sd, "user_tower", list(user), nnum, "cpu")
    assert (p.n_feat, p.emb_dim, p.n_num, p.n_layers) == (6, 16, 13, 3)
    assert list(p.dims[:4]) == [6 * 16 + 13, 512, 256, 256] and list(p.ldw[:3]) == [128, 512, 256]
    off = _arr(pk, p.table_off, (6,), np.int32)
    cards = _arr(pk, p.cards, (6,), np.int32)
    assert list(cards) == list(user.values()) and list(off) == list(np.cumsum([0] + list(user.values())[:-1]))
    tables = _arr(pk, p.tables, (sum(user.values()), 16))
    ucat, unum = synth.user_batch(user, nnum, 37, seed=3)
    x = np.concatenate([tables[off[f] + ucat[:, f]] for f in range(6)] + [unum], axis=1)
    for l in range(3):
        w = _arr(pk, p.w[l], (p.dims[l + 1], p.ldw[l]))
        b = _arr(pk, p.b[l], (p.dims[l + 1],))
        assert not w[:, p.dims[l]:].any()                     # K padding is zero
        x = x @ w[:, :p.dims[l]].T + b
        if l < 2:
            x = np.maximum(x, 0)
    got = oracle.towers.l2_normalize(x.astype(np.float32))
    assert np.abs(got - oracle.towers.user_tower(sd, ucat, unum)).max() <= cases.EMB_ATOL


@pytest.mark.parametrize("fuse", [True, False])
def test_ranker_packing_matches_oracle(fuse):
    user, ad, nnum, sd, _ = cases.ranker_case("ragged", "scaled")
    p, pk, tasks = weights.pack_ranker(sd, list(user), list(ad), nnum, "cpu", fuse_attention=fuse)
    assert tasks == ["ctr", "engagement", "revenue"] and (p.n_layers, p.n_cross, p.n_tasks) == (3, 3, 3)
    assert (p.d_model, p.d_ff, p.head_h1, p.head_h2, p.ldw_proj) == (256, 1024, 256, 64, 864)
    dm = 256
    ucat, unum = synth.user_batch(user, nnum, 21, seed=4)
    acat = synth.ad_features(ad, 21, seed=5)
    feats = oracle.ranker.embed_features(sd, ucat, acat, unum)
    x = feats @ _arr(pk, p.w_proj, (dm, 864))[:, :845].T + _arr(pk, p.b_proj, (dm,))      # pos[0] folded in
    for l in range(3):
        L = p.layers[l]
        if fuse:
            assert not L.w_v
            a = x @ _arr(pk, L.w_o, (dm, dm)).T + _arr(pk, L.b_o, (dm,))
        else:
            a = (x @ _arr(pk, L.w_v, (dm, dm)).T + _arr(pk, L.b_v, (dm,))) @ _arr(pk, L.w_o, (dm, dm)).T \
                + _arr(pk, L.b_o, (dm,))
        x = oracle.ranker.layer_norm(x + a, _arr(pk, L.ln1_g, (dm,)), _arr(pk, L.ln1_b, (dm,)))
        h = np.maximum(x @ _arr(pk, L.w_1, (1024, dm)).T + _arr(pk, L.b_1, (1024,)), 0)
        x = oracle.ranker.layer_norm(x + h @ _arr(pk, L.w_2, (dm, 1024)).T + _arr(pk, L.b_2, (dm,)),
                                     _arr(pk, L.ln2_g, (dm,)), _arr(pk, L.ln2_b, (dm,)))
    x0, xl = x, x
    for c in range(3):
        xl = x0 * (xl @ _arr(pk, p.cross_wt[c], (dm, dm)).T + _arr(pk, p.cross_b[c], (dm,))) + xl   # stored transposed
    h1 = np.maximum(xl @ _arr(pk, p.head_w1, (768, dm)).T + _arr(pk, p.head_b1, (768,)), 0)
    ref = oracle.ranker.forward(sd, ucat, acat, unum)
    scale = cases.logit_scale(ref)
    for t, task in enumerate(tasks):
        h2 = np.maximum(h1[:, 256 * t:256 * (t + 1)] @ _arr(pk, p.head_w2[t], (64, 256)).T
                        + _arr(pk, p.head_b2[t], (64,)), 0)
        logit = h2 @ _arr(pk, p.head_w3[t], (64,)) + _arr(pk, p.head_b3[t], (1,))[0]
        ok, err = cases.logit_close(logit, ref[task], "scaled", scale=scale)
        assert ok, (task, err)


def test_packing_rejects_unsupported_embedding_dim():
    user, ad, nnum = synth.demo_dims()
    sd = synth.two_tower_state(user, ad, nnum, seed=1, embedding_dim=12)
    with pytest.raises(ValueError):
        weights.pack_tower(sd, "user_tower", list(user), nnum, "cpu")


def test_synthetic_criteo_and_preprocessor(tmp_path):
    num, cat, labels = prep.synthetic_criteo(4000)
    assert num.shape == (4000, 13) and len(cat) == 26 and set(np.unique(labels)) <= {0, 1}
    num2, cat2, _ = prep.synthetic_criteo(4000)
    assert np.array_equal(num, num2) and all(np.array_equal(cat[c], cat2[c]) for c in cat)   # seeded (seed 42)
    pp, num_scaled, enc = prep.fit_preprocessor(num, cat)
    assert num_scaled.dtype == np.float32 and np.abs(num_scaled.mean(axis=0)).max() < 1e-4
    assert np.abs(num_scaled.std(axis=0) - 1).max() < 1e-3
    assert "rare" in pp.classes["C1"]                         # 1000 categories over 4000 rows: most are < 10
    assert pp.feature_dims["C26"] == 10 and enc[:, 25].max() == 9
    assert all(enc[:, i].max() < pp.feature_dims[f"C{i + 1}"] for i in range(26))
    # unknown -> 'rare' when the encoder has it, else class 0 (documented deviation from inference.py:180)
    assert pp.encode("C1", "never-seen") == pp.classes["C1"].index("rare")
    assert pp.encode("C26", "never-seen") == 0
    pp.save(tmp_path / "pp.json")
    assert json.load(open(tmp_path / "pp.json"))["numerical_cols"] == [f"I{i}" for i in range(1, 14)]
    pp2 = Preprocessor.load(tmp_path / "pp.json")
    assert pp2.feature_dims == pp.feature_dims and np.allclose(pp2.mean, pp.mean)


def test_struct_sizes_are_stable():
    assert ctypes.sizeof(weights.TowerParams) % 8 == 0 and ctypes.sizeof(weights.RankerParams) % 8 == 0


def test_split_planes_is_an_exact_three_way_bf16_split_in_kernel_layout():
    """weights.split_planes (the host side of the x6 GEMM): h + m + l == w exactly, each plane is a bf16 (low 16 bits
    of its fp32 pattern are zero), layout [out][ld/16][3][16]."""
    from amdrec.weights import split_planes
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((40, 64)) * np.exp(rng.uniform(-20, 20, (40, 64)))).astype(np.float32)
    w[0, :4] = [0.0, -0.0, 1.0, -3.5]
    pl = split_planes(w).view(np.uint16)
    assert pl.shape == (40, 4, 3, 16)
    back = (pl.astype(np.uint32) << 16).view(np.float32)                    # every plane as fp32
    h, m, l = (back[:, :, i, :].reshape(40, 64) for i in range(3))
    assert np.array_equal(h + m + l, w)                                     # exact (h + m is exact, + l is exact)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), w)
    # magnitudes shrink by >= 2^-8 per plane (8 significand bits each)
    nz = w != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -7) and np.all(np.abs(l[nz]) <= np.abs(h[nz]) * 2.0 ** -15)


def test_index_id_codec_round_trips_types():
    """FAISSIndex.save/load keep arbitrary ad ids as what they were (faiss_retrieval.py:208-218 pickles id_map;
    this build's JSON form is typed) and refuse what JSON cannot carry."""
    from amdrec import index
    ids = [17, "ad-3", 2.5, None, np.int64(9), "17"]
    enc = json.loads(json.dumps([index._encode_id(x) for x in ids]))
    dec = [index._decode_id(e) for e in enc]
    assert dec == [17, "ad-3", 2.5, None, 9, "17"] and [type(d) for d in dec] == [int, str, float, type(None), int, str]
    assert index._decode_id("legacy") == "legacy"                 # round-1 files stored str(id)
    for bad in ((1, 2), b"x", True, object()):
        with pytest.raises(TypeError):
            index._encode_id(bad)


def test_grouped_ivf_chunk_limit_holds_for_every_tile_the_scan_may_pick():
    """the query chunk of the grouped IVF scan was sized with the whole batch's tile; a phase / tail chunk
    that falls under the sparse threshold re-picks the smaller tile and overran the launch's query-tile limit for
    nlist above ~37k.  The limit now assumes the smallest tile."""
    from amdrec import ivf
    for nlist in (100, 4096, 37_000, 50_000, 65_000, 65_534):
        for nprobe in (1, 10, 64, 256):
            m = ivf.grouped_chunk_limit(nlist, nprobe)
            assert m >= 1
            for qt in (ivf.QTILE, ivf.QTILE_SPARSE):
                for ncol in {1, max(1, nprobe // 8), nprobe - max(1, nprobe // 8) or 1, nprobe}:
                    assert (m * ncol) // qt + nlist <= ivf.MAX_QUERY_TILES or m == 1
    with pytest.raises(ValueError):
        ivf.grouped_chunk_limit(65_535, 8)


def test_stream_ptr_device_scope_is_restored_by_check(monkeypatch):
    """a call on a tensor that lives on another device must not change the caller's current device.  The
    binding switches in stream_ptr() (last argument of the C call) and switches back in check() (right after it)."""
    import torch
    from amdrec import _lib
    state = {"cur": 0, "log": []}
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["cur"])

    def set_device(d):
        state["cur"] = torch.device(d).index if not isinstance(d, int) else d
        state["log"].append(state["cur"])
    monkeypatch.setattr(torch.cuda, "set_device", set_device)

    asked = []
    monkeypatch.setattr(_lib, "_raw_stream", lambda idx: asked.append(idx) or 1234)   # (the C-side raw-stream getter)
    p = _lib.stream_ptr(torch.device("cuda", 3))
    assert p.value == 1234 and state["cur"] == 3 and asked == [3]
    _lib.check(0)
    assert state["cur"] == 0 and state["log"] == [3, 0]
    _lib.stream_ptr(torch.device("cuda", 0))                 # already current: no switch, nothing to restore
    _lib.check(0)
    assert state["log"] == [3, 0]


def test_ivf_prefilter_choice_by_first_phase_rows_per_result(monkeypatch):
    """amdrec.ivf.use_mixed_scan: the bf16-prefiltered second phase only where the first phase's k-th score is selective
    (configs[4] on one GPU yes; a rank of 8 over the same index and the 1M shapes no); the
    environment forces it for A/B runs."""
    from amdrec import ivf
    monkeypatch.delenv("AMDREC_IVF_MIXED", raising=False)
    monkeypatch.delenv("AMDREC_IVF_FIRST_DIV", raising=False)
    assert ivf.first_phase_probes(64) == 8 and ivf.first_phase_probes(10) == 2
    assert ivf.use_mixed_scan(10_000_000, 4096, 64, 500)           # 8 x 2441 rows for 500 results
    assert not ivf.use_mixed_scan(1_250_000, 4096, 64, 128)        # 8 x 305 rows for 128
    assert not ivf.use_mixed_scan(1_000_000, 4096, 64, 500) and not ivf.use_mixed_scan(1_000_000, 1024, 32, 500)
    monkeypatch.setenv("AMDREC_IVF_MIXED", "1")
    assert ivf.use_mixed_scan(1_000_000, 4096, 64, 500)
    monkeypatch.setenv("AMDREC_IVF_MIXED", "0")
    assert not ivf.use_mixed_scan(10_000_000, 4096, 64, 500)
    monkeypatch.delenv("AMDREC_IVF_MIXED")
    monkeypatch.setenv("AMDREC_IVF_FIRST_DIV", "16")
    assert ivf.first_phase_probes(64) == 4


def test_ivf_scan_choice_by_pairs_per_list():
    """amdrec.ivf.use_grouped_scan: list-major only for batches whose (query, probe) pairs share lists (>= 3 per list on
    average, >= 16 queries); thinly shared or short lists take the pair scan, which runs at the HBM rate of its bytes."""
    from amdrec import ivf
    assert ivf.use_grouped_scan(512, 64, 4096)            # configs[4]: 8 pairs per list
    assert ivf.use_grouped_scan(64, 10, 100)              # the reference's default index, 64 requests: 6.4 per list
    assert not ivf.use_grouped_scan(64, 64, 4096)         # 4096 pairs on up to 4096 lists
    assert not ivf.use_grouped_scan(8, 10, 100)           # below GROUPED_MIN_QUERIES
    assert not ivf.use_grouped_scan(1, 10, 100)
    assert ivf.use_grouped_scan(16, 16, 37)


def test_a_replaced_parameter_object_changes_the_packing_cache_key():
    """the drop-ins key their packed weights on the tensors' `_version`s, read from a cached tensor list; a
    Parameter REPLACED without `invalidate()` used to leave the stale list in place for ever (in-place updates of the new
    tensor never reached the key).  The global registration epoch re-walks the list."""
    from amdrec import _lib
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    k0 = _lib.tensor_versions(m)
    assert _lib.tensor_versions(m) == k0                      # stable while nothing changes
    with torch.no_grad():
        m[0].weight.add_(1.0)
    k1 = _lib.tensor_versions(m)
    assert k1 != k0                                           # in-place update: version bump
    m[0].weight = torch.nn.Parameter(torch.zeros(4, 4))       # replaced object, no invalidate()
    k2 = _lib.tensor_versions(m)
    assert k2 != k1
    with torch.no_grad():
        m[0].weight.add_(1.0)                                 # ... and updates of the NEW tensor are seen
    k3 = _lib.tensor_versions(m)
    assert k3 != k2
    m[1].running_mean = torch.ones(4)                         # a replaced buffer
    assert _lib.tensor_versions(m) != k3
    sd = {k: v.clone() + 1 for k, v in m.state_dict().items()}
    k4 = _lib.tensor_versions(m)
    m[0].load_state_dict({"weight": sd["0.weight"], "bias": sd["0.bias"]}, assign=True)   # assign=True on a child
    assert _lib.tensor_versions(m) != k4


def test_row_owner_engine_eligibility_names_its_reason():
    """the fallback from the f16x3 engine to the generic tile GEMMs must be visible.  The reference's default
    architecture is eligible; the tutorial's (d_model 128, tutorial.ipynb cell 19), an odd d_ff and unfused attention are not,
    each with its reason."""
    from amdrec import weights
    from tests import cases
    user, ad, nnum, sd, _ = cases.ranker_case("demo", "scaled")
    assert weights.x3_ineligible_reason(sd, True) is None and weights.x3_eligible(sd, True)
    assert "fuse_attention" in weights.x3_ineligible_reason(sd, False)
    _, _, _, sd_t, _ = cases.ranker_case("tutorial", "scaled")
    assert weights.x3_ineligible_reason(sd_t, True) == "d_model 128 != 256" and not weights.x3_eligible(sd_t, True)
    sd_ff = synth.ranker_state(user, ad, nnum, seed=3, d_ff=1000)
    assert weights.x3_ineligible_reason(sd_ff, True) == "d_ff 1000 is not a multiple of 32"
    sd_big = synth.ranker_state(user, ad, nnum, seed=3, num_layers=6, d_ff=2048)       # parameter blob beyond the LDS area
    assert "LDS parameter area" in weights.x3_ineligible_reason(sd_big, True)
