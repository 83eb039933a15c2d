"""Host-side weight preparation for libamdrec: fold eval-mode BatchNorm into the preceding
Linear, zero-pad K to a multiple of 32, stack embedding tables, transpose cross weights,
fold the positional row into the projection bias; build the ctypes parameter structs of
include/amdrec.h.  Runs once per weight load (float64 on the host, rounded to float32)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch

MAX_LAYERS = 8
MAX_TASKS = 4
_FP = C.c_void_p
TASKS = ("ctr", "engagement", "revenue")


class TowerParams(C.Structure):
    _fields_ = [("n_feat", C.c_int32), ("emb_dim", C.c_int32), ("n_num", C.c_int32), ("n_layers", C.c_int32),
                ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("ldw", C.c_int32 * MAX_LAYERS),
                ("tables", _FP), ("table_off", _FP), ("cards", _FP),
                ("w", _FP * MAX_LAYERS), ("b", _FP * MAX_LAYERS), ("renormalize", C.c_int32)]


class EncoderLayer(C.Structure):
    _fields_ = [(n, _FP) for n in ("w_v", "b_v", "w_o", "b_o", "ln1_g", "ln1_b", "w_1", "b_1", "w_2", "b_2",
                                   "ln2_g", "ln2_b")] + [("ldw_dm", C.c_int32), ("ldw_ff", C.c_int32)] + \
               [(n, _FP) for n in ("w_o_x6", "w_1_x6", "w_2_x6")]


class X3Weights(C.Structure):
    _fields_ = [("stream", _FP), ("chunks", C.c_int64), ("min_rows", C.c_int64), ("params", _FP), ("n_params", C.c_int64),
                ("variant", C.c_int64),
                ("sw_ov", C.c_float * MAX_LAYERS), ("sw_1", C.c_float * MAX_LAYERS), ("sw_2", C.c_float * MAX_LAYERS),
                ("hn", C.c_float * MAX_LAYERS), ("hb", C.c_float * MAX_LAYERS), ("sw_cross", C.c_float * MAX_LAYERS),
                ("sw_h1", C.c_float), ("sw_h2", C.c_float), ("hn_head", C.c_float), ("hb_head", C.c_float),
                ("stream_cs", _FP), ("chunks_cs", C.c_int64), ("cs_max_rows", C.c_int64)]


class RankerParams(C.Structure):
    _fields_ = [("n_user_feat", C.c_int32), ("n_ad_feat", C.c_int32), ("emb_dim", C.c_int32), ("n_num", C.c_int32),
                ("d_model", C.c_int32), ("d_ff", C.c_int32), ("n_layers", C.c_int32), ("n_cross", C.c_int32),
                ("n_tasks", C.c_int32), ("head_h1", C.c_int32), ("head_h2", C.c_int32), ("ln_eps", C.c_float),
                ("ldw_proj", C.c_int32), ("ldw_cross", C.c_int32), ("ldw_head1", C.c_int32),
                ("ldw_head2", C.c_int32),
                ("tables", _FP), ("table_off", _FP), ("cards", _FP), ("w_proj", _FP), ("b_proj", _FP),
                ("w_proj_user", _FP), ("w_proj_ad", _FP), ("ldw_proj_user", C.c_int32), ("ldw_proj_ad", C.c_int32),
                ("layers", EncoderLayer * MAX_LAYERS),
                ("cross_wt", _FP * MAX_LAYERS), ("cross_b", _FP * MAX_LAYERS),
                ("head_w1", _FP), ("head_b1", _FP),
                ("head_w2", _FP * MAX_TASKS), ("head_b2", _FP * MAX_TASKS),
                ("head_w3", _FP * MAX_TASKS), ("head_b3", _FP * MAX_TASKS),
                ("ad_proj_cache", _FP), ("ld_ad_proj_cache", C.c_int64),
                ("cross_wt_x6", _FP * MAX_LAYERS), ("head_w1_x6", _FP), ("x3", X3Weights)]


def _np64(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().double().numpy()
    return np.asarray(t, dtype=np.float64)


def split_planes(w32: np.ndarray) -> np.ndarray:
    """fp32 [out][ld] (ld % 32 == 0) -> int16 view of [out][ld/16][3][16]: the bf16 planes h, m, l of the exact 3-way
    truncation split w = h + m + l consumed by the x6 GEMM (include/amdrec.h, amdrec_encoder_layer)."""
    w = np.ascontiguousarray(w32, dtype=np.float32)
    out_f, ld = w.shape
    assert ld % 32 == 0
    u = w.view(np.uint32)
    h = (u & np.uint32(0xffff0000)).view(np.float32)
    r1 = w - h                                              # exact
    m = (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    r2 = r1 - m                                             # exact, <= 8 significant bits
    planes = np.stack([(x.view(np.uint32) >> np.uint32(16)).astype(np.uint16) for x in (h, m, r2)], axis=0)
    assert np.array_equal(h + m + r2, w)                    # the split is exact
    return np.ascontiguousarray(planes.reshape(3, out_f, ld // 16, 16).transpose(1, 2, 0, 3)).view(np.int16)


# ---- fp16x3 row-owner engine: fragment packing (mirrors csrc/rowowner.hpp) --------------------------------------------
X3_TARGET_EXP = 12          # scaled maxima lie in [2^12, 2^13)
# position i = 8 h + j of a k-step holds feature offset 8 (j >> 2) + 4 h + (j & 3): bits 2 and 3 of the index swap
_X3_KSRC = np.array([(i & 3) | (((i >> 3) & 1) << 2) | (((i >> 2) & 1) << 3) for i in range(16)])


def x3_pow2_scale(maxabs: float, target_exp: int = X3_TARGET_EXP) -> float:
    """The power of two s with maxabs * s in [2^target_exp, 2^(target_exp+1)) (1.0 for an all-zero matrix)."""
    if not np.isfinite(maxabs) or maxabs <= 0:
        return 1.0
    return float(2.0 ** (target_exp - int(np.floor(np.log2(maxabs)))))


def x3_frags(w64: np.ndarray, scale: float) -> np.ndarray:
    """[N][K] (N % 32 == 0, K % 16 == 0) -> uint16 [N/32][K/16][2 planes][64 lanes][8]: the fp16 planes h = RN16(w s),
    l = RN16(w s - h) of every (32-feature tile, 16-wide k-step) as MFMA A fragments in lane order (lane = p + 32 half
    holds k positions 8 half .. 8 half + 7 of row p), with the accumulator-as-operand k permutation applied."""
    n, k = w64.shape
    assert n % 32 == 0 and k % 16 == 0
    ws = (np.asarray(w64, dtype=np.float64) * scale).astype(np.float32)        # exact: power-of-two scaling of fp32 values
    h = ws.astype(np.float16)
    assert np.isfinite(h).all(), "x3 weight plane overflowed fp16"
    l = (ws - h.astype(np.float32)).astype(np.float16)
    planes = np.stack([h, l]).view(np.uint16)                                  # [2][N][K]
    planes = planes.reshape(2, n, k // 16, 16)[:, :, :, _X3_KSRC]              # position i <- source k offset _X3_KSRC[i]
    planes = planes.reshape(2, n // 32, 32, k // 16, 2, 8)                     # plane, tile, p, ks, half, j
    return np.ascontiguousarray(planes.transpose(1, 3, 0, 4, 2, 5)).reshape(n // 32, k // 16, 2, 64, 8)


def x3_stream_gemm256(fr: np.ndarray) -> np.ndarray:
    """fragments of a [256][256] matrix -> stream order `for ks: for tile: (h, l)` (rowowner.hpp gemm256)."""
    return np.ascontiguousarray(fr.transpose(1, 0, 2, 3, 4)).reshape(-1, 64, 8)


def x3_stream_ffn(f1: np.ndarray, f2: np.ndarray) -> np.ndarray:
    """f1 = fragments of W_1 [d_ff][256], f2 = of W_2 [256][d_ff] -> rowowner.hpp ffn_step order: step t = 0 .. T:
    per micro-step u = 0..15: stage 1 (tile t, ks u) if t < T, then stage 2 (tile u & 7, ks 2 (t - 1) + (u >> 3)) if t >= 1."""
    T = f1.shape[0]
    out = []
    for t in range(T + 1):
        for u in range(16):
            if t < T:
                out += [f1[t, u, 0], f1[t, u, 1]]
            if t >= 1:
                i, ks = u & 7, 2 * (t - 1) + (u >> 3)
                out += [f2[i, ks, 0], f2[i, ks, 1]]
    return np.stack(out)


def x3_stream_heads(f1: np.ndarray, f2s: List[np.ndarray], tiles_per_task: int) -> np.ndarray:
    """f1 = fragments of the stacked head layer 1 [n_tasks * h1][256]; f2s[t] = of head t's layer 2 [64][h1] ->
    rowowner.hpp phase_heads order: per task, per hidden tile: 16 x (h, l) of stage 1, then for s in 0, 1: for i in 0, 1: (h, l)."""
    out = []
    for task, f2 in enumerate(f2s):
        for t in range(tiles_per_task):
            for u in range(16):
                out += [f1[task * tiles_per_task + t, u, 0], f1[task * tiles_per_task + t, u, 1]]
            for sp in range(2):
                for i in range(2):
                    out += [f2[i, 2 * t + sp, 0], f2[i, 2 * t + sp, 1]]
    return np.stack(out)


# ---- 16-rows-per-wave variant (csrc/rowowner16.hpp): 16-feature output tiles, 32-wide k-steps ------------------------
# k position i = 8 g + j of a 32-wide k-step holds feature offset 16 (j >> 2) + 4 g + (j & 3)
_X3B_KSRC = np.array([16 * ((i & 7) >> 2) + 4 * (i >> 3) + (i & 3) for i in range(32)])


def x3b_frags(w64: np.ndarray, scale: float) -> np.ndarray:
    """[N][K] (N % 16 == 0, K % 32 == 0) -> uint16 [N/16][K/32][2 planes][64 lanes][8]: A fragments of
    v_mfma_f32_16x16x32_f16 (lane = p + 16 g holds k positions 8 g .. 8 g + 7 of row p) with the 16-row kernel's k permutation."""
    n, k = w64.shape
    assert n % 16 == 0 and k % 32 == 0
    ws = (np.asarray(w64, dtype=np.float64) * scale).astype(np.float32)
    h = ws.astype(np.float16)
    assert np.isfinite(h).all(), "x3 weight plane overflowed fp16"
    l = (ws - h.astype(np.float32)).astype(np.float16)
    planes = np.stack([h, l]).view(np.uint16)                                  # [2][N][K]
    planes = planes.reshape(2, n, k // 32, 32)[:, :, :, _X3B_KSRC]             # position i <- source k offset
    planes = planes.reshape(2, n // 16, 16, k // 32, 4, 8)                     # plane, tile, p, ks, g, j
    return np.ascontiguousarray(planes.transpose(1, 3, 0, 4, 2, 5)).reshape(n // 16, k // 32, 2, 64, 8)


def _pair(f, t0, ks):
    return [f[t0, ks, 0], f[t0, ks, 1], f[t0 + 1, ks, 0], f[t0 + 1, ks, 1]]


def x3b_stream_gemm256(fr):
    out = []
    for ks in range(fr.shape[1]):
        for tp in range(fr.shape[0] // 2):
            out += _pair(fr, 2 * tp, ks)
    return np.stack(out)


def x3b_stream_cross(fr):
    """A cross layer's 256 x 256 GEMM in four QUARTERS of the output features (4 tiles of 16 = 2 tile pairs each, 16 groups =
    4 chunks): the 16-row kernel keeps x0 in registers across the cross layers and can only afford a 16-register
    accumulator beside it (csrc/rowowner16_impl.hpp phase_cross).  Within a quarter: k-step major, as in gemm256."""
    out = []
    for q4 in range(fr.shape[0] // 4):
        for ks in range(fr.shape[1]):
            for pr in range(2):
                out += _pair(fr, 2 * (2 * q4 + pr), ks)
    return np.stack(out)


def x3b_stream_ffn(f1, f2):
    T = f1.shape[0] // 2                     # hidden tiles of 32
    out = []
    for t in range(T + 1):
        for u in range(8):
            if t < T:
                out += _pair(f1, 2 * t, u)
            if t >= 1:
                out += _pair(f2, 2 * u, t - 1)
    return np.stack(out)


def x3b_stream_heads(f1, f2s, tiles_per_task):
    """Heads of the 16-row kernel, software-pipelined like the FFN (csrc/rowowner16_impl.hpp heads_step): step tt = the 8
    stage-1 groups of hidden tile tt (tiles counted through all tasks) with the 2 stage-2 groups of tile tt - 1 behind
    u = 3 and u = 7; step 0 has no stage 2, the last step no stage 1."""
    nt = len(f2s) * tiles_per_task
    out = []
    for tt in range(nt + 1):
        prev = tt - 1
        for u in range(8):
            if tt < nt:
                out += _pair(f1, 2 * tt, u)
            if tt >= 1 and u in (3, 7):
                out += _pair(f2s[prev // tiles_per_task], 2 * (u // 4), prev % tiles_per_task)
    return np.stack(out)


def _wave_chunk(groups):
    """One 16 KB chunk of the column-split kernel: group w (4 fragment sets) for wave w; None = a wave without work (zeros)."""
    out = []
    for g in groups:
        out += g if g is not None else [np.zeros((64, 8), np.uint16)] * 4
    assert len(out) == 16
    return out


def x3c_stream_gemm256(fr):
    """Column-split kernel (csrc/rowowner16c.hpp): wave w owns output tiles 4 w .. 4 w + 3 = tile pairs 2 w, 2 w + 1; per
    k-step two chunks (pair j of every wave)."""
    assert fr.shape[0] == 16
    out = []
    for ks in range(fr.shape[1]):
        for j in range(2):
            out += _wave_chunk([_pair(fr, 2 * (2 * w + j), ks) for w in range(4)])
    return np.stack(out)


def x3c_stream_ffn(f1, f2):
    """FFN in super-steps of four hidden steps (32 hidden units each), one per wave.  Super-step T: for i < 8 the stage-1
    chunk {wave w: W_1 tile pair of hidden step 4 T + w at k-step i} followed by the stage-2 chunk of super-step T - 1
    {wave w: W_2 output pair 2 w + (i & 1) at k-step 4 (T - 1) + (i >> 1)}."""
    T = f1.shape[0] // 2
    assert T % 4 == 0 and f2.shape[0] == 16
    S = T // 4
    out = []
    for s in range(S + 1):
        for i in range(8):
            if s < S:
                out += _wave_chunk([_pair(f1, 2 * (4 * s + w), i) for w in range(4)])
            if s >= 1:
                out += _wave_chunk([_pair(f2, 2 * (2 * w + (i & 1)), 4 * (s - 1) + (i >> 1)) for w in range(4)])
    return np.stack(out)


def x3c_stream_heads(f1, f2s, tiles_per_task):
    """Heads in super-steps of four hidden steps (counted through all tasks; tiles_per_task % 4 == 0 keeps a super-step inside
    one task).  Stage 1 as in the FFN; stage 2 of super-step s - 1 = four chunks (its k-steps, ascending) behind the stage-1
    chunks i = 3 (two) and i = 7 (two): {wave 0: output pair 0, wave 1: pair 1, waves 2 and 3: zeros}."""
    assert tiles_per_task % 4 == 0
    nt = len(f2s) * tiles_per_task
    S = nt // 4
    out = []
    for s in range(S + 1):
        for i in range(8):
            if s < S:
                out += _wave_chunk([_pair(f1, 2 * (4 * s + w), i) for w in range(4)])
            if s >= 1 and (i & 3) == 3:
                first = 4 * (s - 1)
                f2, k0 = f2s[first // tiles_per_task], first % tiles_per_task
                for kk in (2 * (i >> 2), 2 * (i >> 2) + 1):
                    out += _wave_chunk([_pair(f2, 0, k0 + kk), _pair(f2, 2, k0 + kk), None, None])
    return np.stack(out)


def pack_x3_stream(mats: Dict, variant: int = 32) -> Dict:
    """mats: float64 matrices of the chain {"ov": [L x [256][256]], "w1": [L x [d_ff][256]], "b1": [L x [d_ff]], "w2":
    [L x [256][d_ff]], "cross": [C x [256][256] (already [out][in])], "h1": [T*h1][256], "hb1": [T*h1], "h2": [T x
    [64][h1]]} -> {"stream": uint16 [n_frag][64][8], "chunks", scales and hidden bounds} for amdrec_x3_weights."""
    assert variant in (16, 32, "16cs")                  # "16cs": the 16-row fragments in the column-split kernel's order
    frags, s_gemm, s_ffn, s_heads = {32: (x3_frags, x3_stream_gemm256, x3_stream_ffn, x3_stream_heads),
                                     16: (x3b_frags, x3b_stream_gemm256, x3b_stream_ffn, x3b_stream_heads),
                                     "16cs": (x3b_frags, x3c_stream_gemm256, x3c_stream_ffn, x3c_stream_heads)}[variant]
    s_cross = x3b_stream_cross if variant == 16 else s_gemm
    parts = []
    sc = {"sw_ov": [], "sw_1": [], "sw_2": [], "hn": [], "hb": [], "sw_cross": []}
    for l in range(len(mats["ov"])):
        s_ov = x3_pow2_scale(np.abs(mats["ov"][l]).max())
        parts.append(s_gemm(frags(mats["ov"][l], s_ov)))
        s1, s2 = x3_pow2_scale(np.abs(mats["w1"][l]).max()), x3_pow2_scale(np.abs(mats["w2"][l]).max())
        parts.append(s_ffn(frags(mats["w1"][l], s1), frags(mats["w2"][l], s2)))
        sc["sw_ov"].append(s_ov); sc["sw_1"].append(s1); sc["sw_2"].append(s2)
        # |relu(w_j . x + b_j)| <= ||w_j||_2 ||x||_2 + |b_j| <= (16 max_j ||w_j||_2) max|x| + max_j |b_j|
        sc["hn"].append(float(16.0 * np.linalg.norm(mats["w1"][l], axis=1).max() * (1 + 1e-6)))
        sc["hb"].append(float(np.abs(mats["b1"][l]).max()))
    for w in mats["cross"]:
        s = x3_pow2_scale(np.abs(w).max())
        parts.append(s_cross(frags(w, s)))
        sc["sw_cross"].append(s)
    sh1 = x3_pow2_scale(np.abs(mats["h1"]).max())
    sh2 = x3_pow2_scale(max(np.abs(w).max() for w in mats["h2"]))
    n_tasks = len(mats["h2"])
    tiles = mats["h1"].shape[0] // n_tasks // 32
    parts.append(s_heads(frags(mats["h1"], sh1), [frags(w, sh2) for w in mats["h2"]], tiles))
    stream = np.concatenate(parts)
    assert stream.shape[0] % 16 == 0
    return {"stream": stream, "chunks": stream.shape[0] // 16, "sw_h1": sh1, "sw_h2": sh2,
            "hn_head": float(16.0 * np.linalg.norm(mats["h1"], axis=1).max() * (1 + 1e-6)),
            "hb_head": float(np.abs(mats["hb1"]).max()), **sc}


X3_PARAM_FLOATS = 11264     # LDS parameter area of the kernel (csrc/rowowner.hpp PARAM_FLOATS)


def pack_x3_params(layers: List[Dict], cross_b: List, head_b1, heads: List[Dict]) -> np.ndarray:
    """The parameter blob of the row-owner kernel (layout: csrc/ranker_x3.hip x3_param_floats): per encoder layer
    [b_ov | gamma1 | beta1 | b_1 | b_2 | gamma2 | beta2], per cross layer its bias, heads [stacked b_1] then per task
    [b_2 (64) | w_3 (64) | b_3 padded to 4]; float32, zero-padded to a multiple of 1024."""
    parts = []
    for L in layers:
        parts += [L["b_ov"], L["g1"], L["be1"], L["b1"], L["b2"], L["g2"], L["be2"]]
    parts += list(cross_b)
    parts.append(head_b1)
    for h in heads:
        parts += [h["b2"], h["w3"], np.concatenate([np.asarray(h["b3"], dtype=np.float64).reshape(-1), np.zeros(3)])]
    blob = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in parts]).astype(np.float32)
    n = (len(blob) + 1023) // 1024 * 1024
    out = np.zeros(n, dtype=np.float32)
    out[:len(blob)] = blob
    return out


def _pad_k(w64, mult=32):
    out_f, k = w64.shape
    ld = (k + mult - 1) // mult * mult
    w = np.zeros((out_f, ld), dtype=np.float32)
    w[:, :k] = w64.astype(np.float32)
    return w, ld


class Packed:
    """Keeps the device tensors alive and hands out pointers."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._keep: List[torch.Tensor] = []

    def dev(self, arr, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        t = t.to(self.device)
        assert t.data_ptr() % 16 == 0
        self._keep.append(t)
        return t

    def ptr(self, arr, dtype=None):
        return C.c_void_p(self.dev(arr, dtype).data_ptr())


def pack_tables(pk: Packed, tables: List, emb_dim: int):
    cards = [int(t.shape[0]) for t in tables]
    off = np.concatenate([[0], np.cumsum(cards)[:-1]]).astype(np.int32)
    cat = np.concatenate([_np64(t).astype(np.float32) for t in tables], axis=0)
    assert cat.shape[1] == emb_dim
    return pk.ptr(cat), pk.ptr(off), pk.ptr(np.asarray(cards, dtype=np.int32)), cards


def pack_tower(sd: Dict, prefix: str, feature_names: List[str], n_num: int, device, bn_eps=1e-5):
    """sd: state_dict-like (torch tensors or numpy) with the reference's key names under
    ``prefix`` ('user_tower' / 'ad_tower').  -> (TowerParams, Packed)"""
    pk = Packed(device)
    tables = [sd[f"{prefix}.embedding_layer.embeddings.{n}.weight"] for n in feature_names]
    emb_dim = int(tables[0].shape[1])
    if emb_dim < 4 or emb_dim & (emb_dim - 1):
        raise ValueError("embedding_dim must be a power of two >= 4 for the fused gather")
    p = TowerParams()
    p.n_feat, p.emb_dim, p.n_num = len(tables), emb_dim, n_num
    p.tables, p.table_off, p.cards, _ = pack_tables(pk, tables, emb_dim)
    p.dims[0] = len(tables) * emb_dim + n_num
    idx, l = 0, 0
    while f"{prefix}.mlp.{idx}.weight" in sd:
        w = _np64(sd[f"{prefix}.mlp.{idx}.weight"])
        b = _np64(sd[f"{prefix}.mlp.{idx}.bias"])
        if f"{prefix}.mlp.{idx + 1}.running_mean" in sd:      # Linear followed by BatchNorm1d (eval)
            g = _np64(sd[f"{prefix}.mlp.{idx + 1}.weight"])
            be = _np64(sd[f"{prefix}.mlp.{idx + 1}.bias"])
            mu = _np64(sd[f"{prefix}.mlp.{idx + 1}.running_mean"])
            var = _np64(sd[f"{prefix}.mlp.{idx + 1}.running_var"])
            s = g / np.sqrt(var + bn_eps)
            w = w * s[:, None]
            b = (b - mu) * s + be
            idx += 4
        else:
            idx += 1
        if l >= MAX_LAYERS:
            raise ValueError("too many layers")
        if w.shape[0] % 4:
            raise ValueError("layer widths must be multiples of 4")
        wp, ld = _pad_k(w)
        p.w[l], p.b[l], p.ldw[l], p.dims[l + 1] = pk.ptr(wp), pk.ptr(b.astype(np.float32)), ld, w.shape[0]
        l += 1
    p.n_layers = l
    return p, pk


def x3_ineligible_reason(sd: Dict, fuse_attention: bool):
    """None if the row-owner engine (f16x3) can run this state dict - the architecture it is written for is the
    reference's default: d_model 256, heads 256 -> 64 -> 1 (transformer_ranker.py:213-224, :277-305) - else a short
    reason.  Other architectures (e.g. tutorial.ipynb cell 19: d_model 128) run the generic tile GEMMs."""
    if not fuse_attention:
        return "fuse_attention is off (the engine needs the pre-multiplied W_ov)"
    if "feature_projection.weight" not in sd:
        return "no feature_projection in the state dict"
    dm = int(sd["feature_projection.weight"].shape[0])
    if dm != 256:
        return f"d_model {dm} != 256"
    l = 0
    while f"transformer_layers.{l}.norm1.weight" in sd:
        dff = int(sd[f"transformer_layers.{l}.feed_forward.fc1.weight"].shape[0])
        if dff % 32:
            return f"d_ff {dff} is not a multiple of 32"
        l += 1
    c = 0
    while f"feature_interaction.cross_weights.{c}" in sd:
        c += 1
    tasks = [t for t in TASKS if f"prediction_heads.{t}.0.weight" in sd]
    if not tasks or len(tasks) > MAX_TASKS:
        return f"{len(tasks)} prediction heads (1 .. {MAX_TASKS} supported)"
    if 2 * l + c + 1 > 20:
        return f"{2 * l + c + 1} phases (at most 20)"
    h1 = int(sd[f"prediction_heads.{tasks[0]}.0.weight"].shape[0])
    h2 = int(sd[f"prediction_heads.{tasks[0]}.3.weight"].shape[0])
    if h1 % 32 or h2 != 64:
        return f"head widths {h1} -> {h2} (multiple of 32 -> 64 supported)"
    dff = [int(sd[f"transformer_layers.{i}.feed_forward.fc1.weight"].shape[0]) for i in range(l)]
    n_par = sum(6 * 256 + d for d in dff) + 256 * c + len(tasks) * (h1 + 132)
    if (n_par + 1023) // 1024 * 1024 > X3_PARAM_FLOATS:
        return f"{n_par} bias / LayerNorm parameters exceed the kernel's LDS parameter area ({X3_PARAM_FLOATS})"
    return None


def x3_eligible(sd: Dict, fuse_attention: bool) -> bool:
    return x3_ineligible_reason(sd, fuse_attention) is None


def pack_ranker(sd: Dict, user_names: List[str], ad_names: List[str], n_num: int, device, ln_eps=1e-5,
                fuse_attention: bool = True, x6: bool = True, x3: bool = False, x3_min_rows: int = 0,
                x3_variant: int = 32, x3_cs_max_rows: int = 0):
    """state_dict-like of the reference TransformerRanker -> (RankerParams, Packed, task names).
    ``fuse_attention``: pre-multiply W_ov = W_o W_v, b_ov = W_o b_v + b_o in float64 (the seq-len-1
    attention is exactly W_o(W_v x + b_v) + b_o, transformer_ranker.py:59-88 with :358), so each
    encoder layer's attention block is one GEMM instead of two.
    ``x6``: also upload the bf16 split planes of the big weight matrices (W_ov / W_o, fc1, fc2, cross, stacked head
    layer 1) so that passes of more than 8192 rows run on the error-compensated bf16-MFMA GEMM."""
    pk = Packed(device)
    tables = [sd[f"user_embeddings.{n}.weight"] for n in user_names] + \
             [sd[f"ad_embeddings.{n}.weight"] for n in ad_names]
    emb_dim = int(tables[0].shape[1])
    if emb_dim < 4 or emb_dim & (emb_dim - 1):
        raise ValueError("embedding_dim must be a power of two >= 4 for the fused gather")
    p = RankerParams()
    p.n_user_feat, p.n_ad_feat, p.emb_dim, p.n_num = len(user_names), len(ad_names), emb_dim, n_num
    p.tables, p.table_off, p.cards, _ = pack_tables(pk, tables, emb_dim)
    wproj = _np64(sd["feature_projection.weight"])
    d_model = wproj.shape[0]
    assert wproj.shape[1] == len(tables) * emb_dim + n_num
    p.d_model = d_model
    w, p.ldw_proj = _pad_k(wproj)
    p.w_proj = pk.ptr(w)
    # split for the broadcast form: [user emb | numerical] and [ad emb] column blocks
    nu, na = len(user_names) * emb_dim, len(ad_names) * emb_dim
    if nu + n_num > 0 and na > 0:
        w, p.ldw_proj_user = _pad_k(np.concatenate([wproj[:, :nu], wproj[:, nu + na:]], axis=1))
        p.w_proj_user = pk.ptr(w)
        w, p.ldw_proj_ad = _pad_k(wproj[:, nu:nu + na])
        p.w_proj_ad = pk.ptr(w)
    pos0 = _np64(sd["positional_encoding"])[0, 0]                  # only row 0 is ever read (:361)
    p.b_proj = pk.ptr((_np64(sd["feature_projection.bias"]) + pos0).astype(np.float32))
    l = 0
    while f"transformer_layers.{l}.norm1.weight" in sd:
        pre = f"transformer_layers.{l}"
        L = p.layers[l]
        wv, bv = _np64(sd[f"{pre}.self_attention.W_v.weight"]), _np64(sd[f"{pre}.self_attention.W_v.bias"])
        wo, bo = _np64(sd[f"{pre}.self_attention.W_o.weight"]), _np64(sd[f"{pre}.self_attention.W_o.bias"])
        if fuse_attention:
            mats = (("w_o", wo @ wv), ("w_1", _np64(sd[f"{pre}.feed_forward.fc1.weight"])))
            bo = wo @ bv + bo
            L.w_v, L.b_v = None, None
        else:
            mats = (("w_v", wv), ("w_o", wo), ("w_1", _np64(sd[f"{pre}.feed_forward.fc1.weight"])))
            L.b_v = pk.ptr(bv.astype(np.float32))
        for dst, mat in mats:
            w, ld = _pad_k(mat)
            setattr(L, dst, pk.ptr(w))
            L.ldw_dm = ld
            if x6 and dst in ("w_o", "w_1"):
                setattr(L, dst + "_x6", pk.ptr(split_planes(w)))
        L.b_o = pk.ptr(bo.astype(np.float32))
        w, L.ldw_ff = _pad_k(_np64(sd[f"{pre}.feed_forward.fc2.weight"]))
        L.w_2 = pk.ptr(w)
        if x6:
            L.w_2_x6 = pk.ptr(split_planes(w))
        p.d_ff = int(sd[f"{pre}.feed_forward.fc1.weight"].shape[0])
        for dst, src in (("b_1", "feed_forward.fc1.bias"), ("b_2", "feed_forward.fc2.bias"),
                         ("ln1_g", "norm1.weight"), ("ln1_b", "norm1.bias"),
                         ("ln2_g", "norm2.weight"), ("ln2_b", "norm2.bias")):
            setattr(L, dst, pk.ptr(_np64(sd[f"{pre}.{src}"]).astype(np.float32)))
        l += 1
    p.n_layers = l
    if l == 0:
        p.d_ff = 4
    c = 0
    while f"feature_interaction.cross_weights.{c}" in sd:
        w, p.ldw_cross = _pad_k(_np64(sd[f"feature_interaction.cross_weights.{c}"]).T)   # xl @ W == xl (W^T)^T
        p.cross_wt[c] = pk.ptr(w)
        if x6:
            p.cross_wt_x6[c] = pk.ptr(split_planes(w))
        p.cross_b[c] = pk.ptr(_np64(sd[f"feature_interaction.cross_biases.{c}"]).astype(np.float32))
        c += 1
    p.n_cross = c
    tasks = [t for t in TASKS if f"prediction_heads.{t}.0.weight" in sd]
    p.n_tasks = len(tasks)
    w1 = np.concatenate([_np64(sd[f"prediction_heads.{t}.0.weight"]) for t in tasks], axis=0)
    b1 = np.concatenate([_np64(sd[f"prediction_heads.{t}.0.bias"]) for t in tasks], axis=0)
    p.head_h1 = int(sd[f"prediction_heads.{tasks[0]}.0.weight"].shape[0])
    p.head_h2 = int(sd[f"prediction_heads.{tasks[0]}.3.weight"].shape[0])
    w, p.ldw_head1 = _pad_k(w1)
    p.head_w1, p.head_b1 = pk.ptr(w), pk.ptr(b1.astype(np.float32))
    if x6:
        p.head_w1_x6 = pk.ptr(split_planes(w))
    for i, t in enumerate(tasks):
        w, p.ldw_head2 = _pad_k(_np64(sd[f"prediction_heads.{t}.3.weight"]))
        p.head_w2[i] = pk.ptr(w)
        p.head_b2[i] = pk.ptr(_np64(sd[f"prediction_heads.{t}.3.bias"]).astype(np.float32))
        p.head_w3[i] = pk.ptr(_np64(sd[f"prediction_heads.{t}.6.weight"]).reshape(-1).astype(np.float32))
        p.head_b3[i] = pk.ptr(_np64(sd[f"prediction_heads.{t}.6.bias"]).reshape(-1).astype(np.float32))
    p.ln_eps = ln_eps
    if x3 and x3_eligible(sd, fuse_attention):
        # the SAME fp32-rounded matrices the other engines multiply with, split into fp16 planes in stream order
        f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)   # noqa: E731
        mats = {"ov": [], "w1": [], "b1": [], "w2": [], "cross": [], "h2": []}
        for li in range(p.n_layers):
            pre = f"transformer_layers.{li}"
            wv, wo = _np64(sd[f"{pre}.self_attention.W_v.weight"]), _np64(sd[f"{pre}.self_attention.W_o.weight"])
            mats["ov"].append(f32(wo @ wv))
            mats["w1"].append(f32(_np64(sd[f"{pre}.feed_forward.fc1.weight"])))
            mats["b1"].append(f32(_np64(sd[f"{pre}.feed_forward.fc1.bias"])))
            mats["w2"].append(f32(_np64(sd[f"{pre}.feed_forward.fc2.weight"])))
        for ci in range(p.n_cross):
            mats["cross"].append(f32(_np64(sd[f"feature_interaction.cross_weights.{ci}"]).T))
        mats["h1"], mats["hb1"] = f32(w1), f32(b1)
        for t in tasks:
            mats["h2"].append(f32(_np64(sd[f"prediction_heads.{t}.3.weight"])))
        x = pack_x3_stream(mats, x3_variant)
        p.x3.variant = x3_variant
        lay = []
        for li in range(p.n_layers):
            pre = f"transformer_layers.{li}"
            wo = _np64(sd[f"{pre}.self_attention.W_o.weight"])
            lay.append({"b_ov": wo @ _np64(sd[f"{pre}.self_attention.W_v.bias"]) + _np64(sd[f"{pre}.self_attention.W_o.bias"]),
                        "g1": _np64(sd[f"{pre}.norm1.weight"]), "be1": _np64(sd[f"{pre}.norm1.bias"]),
                        "b1": _np64(sd[f"{pre}.feed_forward.fc1.bias"]), "b2": _np64(sd[f"{pre}.feed_forward.fc2.bias"]),
                        "g2": _np64(sd[f"{pre}.norm2.weight"]), "be2": _np64(sd[f"{pre}.norm2.bias"])})
        blob = pack_x3_params(lay, [_np64(sd[f"feature_interaction.cross_biases.{ci}"]) for ci in range(p.n_cross)], b1,
                              [{"b2": _np64(sd[f"prediction_heads.{t}.3.bias"]), "w3": _np64(sd[f"prediction_heads.{t}.6.weight"]),
                                "b3": _np64(sd[f"prediction_heads.{t}.6.bias"])} for t in tasks])
        assert len(blob) <= X3_PARAM_FLOATS
        p.x3.params = pk.ptr(blob)
        p.x3.n_params = len(blob)
        p.x3.stream = pk.ptr(x["stream"].view(np.int16))
        p.x3.chunks = x["chunks"]
        p.x3.min_rows = int(x3_min_rows)
        if x3_variant == 16 and x3_cs_max_rows >= 0 and p.d_ff % 128 == 0 and p.head_h1 % 128 == 0:
            xc = pack_x3_stream(mats, "16cs")               # same planes and scales, the column-split kernel's chunk order
            assert all(xc[k] == x[k] for k in ("sw_ov", "sw_1", "sw_2", "sw_cross", "sw_h1", "sw_h2"))
            p.x3.stream_cs = pk.ptr(xc["stream"].view(np.int16))
            p.x3.chunks_cs = xc["chunks"]
        p.x3.cs_max_rows = int(x3_cs_max_rows)
        for li in range(p.n_layers):
            p.x3.sw_ov[li], p.x3.sw_1[li], p.x3.sw_2[li] = x["sw_ov"][li], x["sw_1"][li], x["sw_2"][li]
            p.x3.hn[li], p.x3.hb[li] = x["hn"][li], x["hb"][li]
        for ci in range(p.n_cross):
            p.x3.sw_cross[ci] = x["sw_cross"][ci]
        p.x3.sw_h1, p.x3.sw_h2, p.x3.hn_head, p.x3.hb_head = x["sw_h1"], x["sw_h2"], x["hn_head"], x["hb_head"]
    return p, pk, tasks
