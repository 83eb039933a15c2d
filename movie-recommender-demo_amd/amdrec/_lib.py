"""ctypes binding of libamdrec.so (include/amdrec.h).

There is no CPU fallback: if the library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMDREC_LIB_PATH: developer override for A/B runs of two builds of the library in otherwise identical processes
LIB_PATH = os.environ.get("AMDREC_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libamdrec.so")

ABI_VERSION = 10
MAX_K = 2048


class AmdrecError(RuntimeError):
    pass


_lib = None

_i64, _i32, _sz, _vp, _fp = C.c_int64, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/amdrec.h
_SIGNATURES = {
    "amdrec_abi_version": [],
    "amdrec_last_error": [],
    "amdrec_flat_search_workspace": [_i64, _i64, _i32, C.POINTER(_sz)],
    "amdrec_bf16_rows": [_fp, _i64, _i64, _i32, _vp, _i64, _fp, _vp],
    "amdrec_flat_search_mixed_workspace": [_i64, _i64, _i32, _i32, C.POINTER(_sz)],
    "amdrec_flat_search_mixed": [_fp, _i64, _i64, _i32, _vp, _i64, _fp, _fp, _i64, _i64, _i32, _i64, _fp, _vp, _vp,
                                 _sz, _vp, _vp],
    "amdrec_flat_search": [_fp, _i64, _i64, _i32, _fp, _i64, _i64, _i32, _i64, _fp, _vp, _vp, _sz, _vp, _vp],
    "amdrec_ivf_scan": [_fp, _i64, _i32, _vp, _vp, _fp, _i64, _i64, _vp, _vp, _i32, _vp, _i64, _i64, _vp],
    "amdrec_ivf_scan_grouped": [_fp, _i64, _i32, _vp, _vp, _i32, _i64, _fp, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32,
                                _vp, _i64, _i64, _fp, _i64, _vp, _vp],
    "amdrec_ivf_filter_bounds": [_fp, _i64, _i64, _i32, _vp, _i64, _fp, _fp, _i64, _fp, _vp],
    "amdrec_ivf_scan_grouped_mixed": [_fp, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _i64, _fp, _i64, _vp, _i64, _vp, _vp, _i64,
                                      _i32, _vp, _vp, _i64, _i64, _fp, _i64, _fp, _vp, _vp],
    "amdrec_ivf_select": [_vp, _i64, _vp, _i64, _i32, _fp, _vp, _vp],
    "amdrec_ivf_select_split": [_vp, _i64, _vp, _i64, _i32, _i32, _fp, _vp, _vp, _sz, _vp, _vp],
    "amdrec_ivf_coarse_keys": [_fp, _i32, _i64, _i32, _fp, _i64, _i64, _vp, _i64, _vp],
    "amdrec_ivf_group": [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp],
    "amdrec_ivf_assign": [_fp, _i64, _i64, _i32, _fp, _i32, _i64, _vp, _fp, _vp, _sz, _vp],
    "amdrec_ivf_kmeans_workspace": [_i64, _i32, _i32, C.POINTER(_sz)],
    "amdrec_ivf_kmeans_step": [_fp, _i64, _i64, _i32, _fp, _i32, _i64, _vp, _sz, _vp],
    "amdrec_topk_merge": [_fp, _vp, _i32, _i64, _i64, _i64, _i32, _fp, _vp, _vp],
    "amdrec_topk_merge_partial": [_fp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _fp, _vp, _vp, _vp],
    "amdrec_tower_workspace": [_vp, _i64, C.POINTER(_sz)],
    "amdrec_tower_forward": [_vp, _vp, _fp, _i64, _fp, _i64, _vp, _vp, _sz, _vp],
    "amdrec_ranker_workspace": [_vp, _i64, C.POINTER(_sz)],
    "amdrec_ranker_forward": [_vp, _vp, _fp, _i64, _vp, _vp, _i64, _fp, _i64, _vp, _i64, _i64, _vp, _sz, _vp],
    "amdrec_l2_normalize": [_fp, _i64, _fp, _i64, _i64, _i32, _vp],
    "amdrec_remap_ids": [_vp, _vp, _i64, _vp, _i64, _vp],
    "amdrec_profile_enable": [_i32],
    "amdrec_profile_only": [C.c_char_p],
    "amdrec_profile_report": [_vp, _i32, C.POINTER(_i32)],
    "amdrec_ranker_project_ads": [_vp, _vp, _i64, _fp, _i64, _vp, _sz, _vp],
    "amdrec_ranker_x3_prefix": [_vp, _fp, _i64, _i64, _i32, _fp, _i64, _fp, _i64, _vp, _sz, _vp],
    "amdrec_prep_numerical": [_fp, _fp, _fp, _fp, _i64, _i32, _vp],
    "amdrec_select_topk": [_fp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _fp, _vp, _vp],
}
_RESTYPE = {"amdrec_last_error": C.c_char_p}


def exported_symbols():
    return list(_SIGNATURES)


def load():
    """Load libamdrec.so once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmdrecError(
            f"{LIB_PATH} not found: build it with `python movie-recommender-demo_amd/build.py` "
            "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    v = lib.amdrec_abi_version()
    if v != ABI_VERSION:
        raise AmdrecError(f"libamdrec ABI {v} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


_tls = threading.local()          # .prev_device: the thread's HIP device before stream_ptr() switched it (else None)


def check(status):
    """Status check of a C call - and the end of the call's device scope: if ``stream_ptr`` switched the calling
    thread's current HIP device for this call, switch it back (a call on a tensor that lives on cuda:N must not leave
    the caller's later ``device='cuda'`` allocations and launches on cuda:N)."""
    prev = getattr(_tls, "prev_device", None)
    if prev is not None:
        _tls.prev_device = None
        torch.cuda.set_device(prev)
    if status != 0:
        msg = load().amdrec_last_error()
        raise AmdrecError(f"libamdrec error {status}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` - and make ``device`` the calling thread's current HIP
    device for the duration of ONE C call: the C entry points launch kernels, set function attributes and memset on the
    CURRENT device (amdrec.h), so a tensor on cuda:N (N != 0), or a call from a fresh thread (whose current device is
    0), must switch first.  Every binding call site evaluates this as the last argument of ``check(lib.fn(...,
    stream_ptr(dev)))``: the switch happens right before the C call and ``check`` restores the previous device right
    after it, so the caller's current device is unchanged by an amdrec call."""
    cur = torch.cuda.current_device()
    idx = cur
    if device is not None:
        dev = device if isinstance(device, torch.device) else torch.device(device)
        if dev.type == "cuda" and dev.index is not None:
            idx = dev.index
            if idx != cur:
                if getattr(_tls, "prev_device", None) is None:
                    _tls.prev_device = cur
                torch.cuda.set_device(dev)
    # the raw handle straight from the C side: torch.cuda.current_stream() builds a Stream object through three layers of
    # device-index parsing - 5 us per call, nine calls per request (a seventh of a single request's host time)
    return C.c_void_p(_raw_stream(idx))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


def require_gpu(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise AmdrecError(f"{name} must be a tensor on a HIP device (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise AmdrecError(f"{name} must be {dtype}, got {t.dtype}")
    return t


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


def profile_enable(on: bool, only: str = ""):
    """Per-launch HIP-event timing on / off (clears the counters); ``only``: time just the tags with this prefix."""
    check(load().amdrec_profile_only(only.encode() if only else None))
    check(load().amdrec_profile_enable(1 if on else 0))


def profile_report():
    """-> {tag: dict(launches, total_ms, flops, bytes)} (synchronises with the recorded events)."""
    arr = (ProfileEntry * 64)()
    n = C.c_int(0)
    check(load().amdrec_profile_report(C.cast(arr, C.c_void_p), 64, C.byref(n)))
    return {arr[i].name.decode(): {"launches": arr[i].launches, "total_ms": arr[i].total_ms,
                                   "flops": arr[i].flops, "bytes": arr[i].bytes} for i in range(n.value)}


# Every registration of a Parameter / buffer / submodule on ANY nn.Module bumps this epoch (torch's global registration
# hooks: `module.weight = nn.Parameter(...)`, `load_state_dict(assign=True)` on a child and parametrize all go through
# register_parameter; assigning to an existing buffer name calls the buffer hooks).  One integer compare per forward tells
# tensor_versions() whether some tensor OBJECT may have been replaced since it cached its list.
_REG_EPOCH = [0]


def _bump_registration_epoch(*_args):
    _REG_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_registration_epoch)
torch.nn.modules.module.register_module_buffer_registration_hook(_bump_registration_epoch)
torch.nn.modules.module.register_module_module_registration_hook(_bump_registration_epoch)


def tensor_versions(module) -> tuple:
    """``_version`` of every parameter and buffer of ``module`` - the part of the drop-ins' packing-cache key that detects
    in-place weight updates (optimizer steps, ``copy_``).  The tensor list itself is cached on the module: walking
    ``parameters()`` on every call cost 0.1 ms per forward on the serving path (a fifth of a single request).  The cache is
    dropped by the drop-ins' ``_apply`` / ``load_state_dict`` / ``invalidate``, and re-walked whenever any Parameter, buffer
    or submodule has been (re-)registered anywhere since it was taken (``_REG_EPOCH``): a REPLACED Parameter object is
    seen - the new tensor's identity enters the key - and the replaced tensors are released."""
    d = module.__dict__
    tl = d.get("_amdrec_tensor_list")
    if tl is None or d.get("_amdrec_tensor_epoch") != _REG_EPOCH[0]:
        tl = [*module.parameters(), *module.buffers()]
        d["_amdrec_tensor_list"] = tl
        d["_amdrec_tensor_epoch"] = _REG_EPOCH[0]
        d["_amdrec_tensor_ids"] = hash(tuple(id(t) for t in tl))
    return (d["_amdrec_tensor_ids"], *[t._version for t in tl])


def drop_tensor_list(module):
    module.__dict__.pop("_amdrec_tensor_list", None)


class Workspace:
    """Grow-only per-device scratch buffer handed to the C ABI (caller-owned workspace).

    Kernels launched on one stream use it one after the other, so a single buffer per device is enough
    for eager calls.  A captured HIP graph bakes the buffer's address into its kernel nodes, so it must
    never share (or outlive) a buffer that may be re-allocated: ``private()`` installs a separate,
    pre-sized arena for the duration of a capture and the graph owner keeps it alive."""

    def __init__(self):
        self._buf = {}
        self._override = None

    def get(self, nbytes, device):
        if self._override is not None:
            return self._override.get(nbytes, device)
        key = torch.device(device)
        if key.index is None and key.type == "cuda":
            key = torch.device("cuda", torch.cuda.current_device())
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._buf[key] = None
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=key)
            assert buf.data_ptr() % 256 == 0
            self._buf[key] = buf
        return buf

    def private(self, arena: "FixedArena"):
        ws = self

        class _Ctx:
            def __enter__(self_):
                self_.prev, ws._override = ws._override, arena
                return arena

            def __exit__(self_, *exc):
                ws._override = self_.prev
                return False
        return _Ctx()


class FixedArena:
    """A workspace of fixed size that refuses to grow (for graph capture)."""

    def __init__(self, nbytes, device):
        self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self.high_water = 0

    def get(self, nbytes, device):
        self.high_water = max(self.high_water, int(nbytes))
        if nbytes > self.buf.numel():
            raise AmdrecError(f"graph workspace too small: need {nbytes} bytes, have {self.buf.numel()}")
        return self.buf


class MeasuringArena:
    """Records the largest request while delegating to the shared workspace (sizing pass)."""

    def __init__(self, inner):
        self.inner, self.high_water = inner, 0

    def get(self, nbytes, device):
        self.high_water = max(self.high_water, int(nbytes))
        return self.inner.get(nbytes, device)


WORKSPACE = Workspace()
