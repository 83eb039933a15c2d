"""Device-side synthetic corpora for the benchmark and the full-size tests (the reference's own
benchmark distribution, faiss_retrieval.py:390: ``randn(N, 256)`` rows, L2-normalised as ``add`` does, :114-115).

Rows are generated on the device in fixed 65536-row blocks, each block from its own seeded generator, so any shard
``[row0, row0 + rows)`` of the same ``(n, seed)`` corpus is bit-identical wherever it is generated (every rank of
the sharded benchmark builds only its own rows; a one-GPU test can cut the same corpus into eight shards)."""
from __future__ import annotations

import torch

BLOCK = 65536


def _blocks(n, row0, rows):
    b0, b1 = row0 // BLOCK, (row0 + rows + BLOCK - 1) // BLOCK
    for b in range(b0, b1):
        s, e = max(b * BLOCK, row0), min((b + 1) * BLOCK, row0 + rows, n)
        if e > s:
            yield b, s, e


def device_corpus(n, dim, device, seed=1234, row0=0, rows=None):
    """randn rows, L2-normalised."""
    rows = n - row0 if rows is None else rows
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    for b, s, e in _blocks(n, row0, rows):
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + b)
        x = torch.randn((BLOCK, dim), generator=g, device=device, dtype=torch.float32)
        x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-30)
        out[s - row0:e - row0] = x[s - b * BLOCK:e - b * BLOCK]
    return out


def device_clustered_corpus(n, dim, device, n_clusters=4096, spread=0.5, seed=1234, row0=0, rows=None):
    """A corpus with cluster structure (what an inverted-file index exists for; on i.i.d. random unit vectors no
    partition of the sphere helps and recall@k of any IVF is ~ the probed fraction): row = centre[c] + spread *
    randn / sqrt(dim) scaled to the centre's norm, L2-normalised; centres are randn unit vectors."""
    rows = n - row0 if rows is None else rows
    g0 = torch.Generator(device=device)
    g0.manual_seed(seed * 7_919 + 17)
    cent = torch.randn((n_clusters, dim), generator=g0, device=device, dtype=torch.float32)
    cent = cent / cent.norm(dim=1, keepdim=True)
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    for b, s, e in _blocks(n, row0, rows):
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + b)
        c = torch.randint(0, n_clusters, (BLOCK,), generator=g, device=device)
        x = cent[c] + spread * torch.randn((BLOCK, dim), generator=g, device=device, dtype=torch.float32) / (dim ** 0.5)
        x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-30)
        out[s - row0:e - row0] = x[s - b * BLOCK:e - b * BLOCK]
    return out
