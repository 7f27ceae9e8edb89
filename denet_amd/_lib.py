"""ctypes binding of libdenet_hip.so (the C ABI declared in include/denet_hip.h).

There is no CPU fallback: if the library is missing this module raises at
import of the first op, and every op raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DENET_HIP_LIB: another build of the same library (kernel A/B experiments); never a different backend
LIB_PATH = os.environ.get("DENET_HIP_LIB") or os.path.join(_HERE, "libdenet_hip.so")

F32, BF16 = 0, 1
BWD_HAVE_FOCAL, BWD_NO_DFS, BWD_ONLY_DFS = 1, 2, 4      # denet_head_step_bwd flags (include/denet_hip.h)

# signature table: p = device pointer of any dtype (features, workspace; tensor / None),
# F = device pointer to float32, L = device pointer to int64 (both checked against the tensor's dtype:
# the kernels read raw memory, so a bf16 / half / int32 tensor would be misread without an error),
# l = int64, u = uint64, i = int, f = float, h = host pointer (ctypes object / None)
# the trailing stream argument is appended automatically ("s").
_SIGS = {
    "denet_mask_resize": "L" "lllll" "FF",
    "denet_masked_sum_fwd": "piF" "lll" "F",
    "denet_masked_sum_bwd": "FF" "lll" "pi",
    "denet_masked_sum_bwd_chunked": "FF" "lll" "pi",
    "denet_masked_pool_fwd": "piL" "llllll" "FFF",
    "denet_proto_extend_fwd": "FFFFFFFFF" "L" "FF" "llllll" "FFFFF",
    "denet_proto_extend_bwd": "Fl" "FFFFF" "FFF" "LFF" "llllll" "FFFFFFFF" "pl",
    "denet_proto_extend_bwd_static": "Fl" "FFFFF" "FFF" "LFF" "llllll" "FFFFFFFF",
    "denet_classify_fwd": "piFL" "lllll" "i" "FF",
    "denet_classify_shared_fwd": "piF" "llll" "F",
    "denet_classify_bwd": "piF" "FFFF" "llll" "pF",
    "denet_binary_logits": "FL" "llll" "i" "F",
    "denet_batch_mean": "F" "ll" "F",
    "denet_sum_parts": "F" "lll" "F",
    "denet_upsample_wce_fwd": "FLF" "llllll" "FFF",
    "denet_upsample_focal_fwd": "FL" "lllll" "ff" "F",
    "denet_loss_finalize": "FF" "llll" "i" "FFF",
    "denet_combine_slabs": "FFFF" "ll" "F",
    "denet_upsample_bilinear": "F" "lllll" "F",
    "denet_confmat": "FL" "lll" "L",
    "denet_confmat_fused": "FL" "llllll" "L",
    "denet_binary_iou_counts": "FL" "ll" "L",
    "denet_binary_iou_counts_fused": "FL" "lllll" "L",
    "denet_head_step_fwd": "ppi" "LLLLF" "FFFFFFF" "FF" "lllllllll" "i" "FFF" "pl" "h",
    "denet_head_step_bwd": "piL" "FFF" "FF" "F" "lllllllll" "pp" "FFFFFFF" "Fi" "pl" "h",
    "denet_head_step_train_fwd": "ppi" "LLLLF" "FFFFFFF" "FF" "h" "lllllllll" "i" "FFF" "pl" "h",
    "denet_head_step_train_bwd": "piL" "FFF" "FF" "h" "F" "lllllllll" "pp" "FFFFFFF" "Fi" "pl" "h",
    "denet_dropout_mask": "uifl" "F",
}
_NOSTREAM = {
    "denet_version": (ctypes.c_char_p, ""),
    "denet_strerror": (ctypes.c_char_p, "i"),
    "denet_proto_extend_bwd_workspace": (ctypes.c_int64, "lllll"),
    "denet_proto_extend_bwd_static_limit": (ctypes.c_int64, ""),
    "denet_classify_bwd_parts": (ctypes.c_int64, "l"),
    "denet_head_step_workspace": (ctypes.c_int64, "lllllllll"),
    "denet_upsample_loss_parts": (ctypes.c_int64, "llllli"),
    "denet_head_step_plan": (ctypes.c_int, "lllllllllih"),
}
_CT = {"p": ctypes.c_void_p, "F": ctypes.c_void_p, "L": ctypes.c_void_p, "l": ctypes.c_int64, "u": ctypes.c_uint64, "i": ctypes.c_int, "f": ctypes.c_float, "h": ctypes.c_void_p}


class Probe(ctypes.Structure):
    """denet_probe_t (include/denet_hip.h)."""
    _fields_ = [("which", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("start", ctypes.c_void_p), ("stop", ctypes.c_void_p)]


class Dropout(ctypes.Structure):
    """denet_dropout_t (include/denet_hip.h): in-kernel train-mode dropout of the fused step.  The library
    reads the struct during each call, so a bound step sees a `seed` changed in place."""
    _fields_ = [("seed", ctypes.c_uint64), ("p_gate", ctypes.c_float), ("p_weight", ctypes.c_float)]


DROPOUT_GATE, DROPOUT_WEIGHT = 1, 2                     # stream ids of denet_dropout_mask


class StepPlan(ctypes.Structure):
    """denet_step_plan_t (include/denet_hip.h): which kernel variant each launch of the fused step runs."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "pool_launches", "pool_long_rows", "pool_nt_loads", "classify_layout", "classify_groups",
        "loss_chain_cell", "loss_chunk", "loss_xsplit", "bwd_persistent", "bwd_groups", "bwd_dfq_nt",
        "dfs_nt", "dfs_flat_writer", "dfs_chunked", "bwd_bf16_pipe", "bwd_xcd_pairs")]


CLS_TILE8, CLS_UNITS, CLS_WAVE4, CLS_WAVE4_BF16 = 0, 1, 2, 3


def head_step_plan(N, way, shot, K, C, h, w, H, W, feat_dtype=None):
    """dict of the launchers' choices for a shape (denet_head_step_plan); feat_dtype: torch dtype or F32 / BF16."""
    code = feat_dtype if isinstance(feat_dtype, int) else (BF16 if feat_dtype == torch.bfloat16 else F32)
    plan = StepPlan()
    rc = load().denet_head_step_plan(N, way, shot, K, C, h, w, H, W, code, ctypes.byref(plan))
    if rc != 0:
        raise DenetHipError(f"denet_head_step_plan failed: {strerror(rc)} (code {rc})")
    return {n: int(getattr(plan, n)) for n, _ in StepPlan._fields_ if n != "reserved"}


# the launches of the fused step (include/denet_hip.h, DENET_K_*)
KERNEL_IDS = {"masked_pool_fwd": 2, "proto_extend_fwd": 3, "classify_fwd": 4, "loss_pair_fwd": 5,
              "loss_finalize": 7, "bwd_dwn": 8, "bwd_chain_dfq": 9, "bwd_params_dfs": 10}
GRAD_DEFER_LOSSES = 2

EXPORTS = sorted(list(_SIGS) + list(_NOSTREAM))

_lib = None


class DenetHipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for loading itself)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DenetHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C denet_amd/csrc`. denet_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [_CT[c] for c in sig] + [ctypes.c_void_p]
    for name, (res, sig) in _NOSTREAM.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = [_CT[c] for c in sig]
    _lib = lib
    return lib


_WANT = {"F": torch.float32, "L": torch.int64}


def _ptr(x, code="p", where=""):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        want = _WANT.get(code)
        if want is not None and x.dtype != want:
            raise DenetHipError(f"{where}: expected a {want} tensor, got {x.dtype} "
                                "(the kernels read raw memory; cast explicitly)")
        if not x.is_cuda:
            raise DenetHipError("denet_amd ops need device tensors (got a CPU tensor); there is no CPU path")
        if not x.is_contiguous():
            raise DenetHipError("denet_amd ops need contiguous tensors")
        return x.data_ptr()
    return int(x)


def strerror(code: int) -> str:
    return load().denet_strerror(code).decode()


def current_stream_handle(device_index=None) -> int:
    """hipStream_t of torch's current stream on the device (default: the current device) as an integer.
    torch.cuda.current_stream() builds a Stream object per call (~8 us); the raw getter is ~0.3 us, which
    matters for entry points that are called twice per 90 us step."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def call(name: str, *args):
    """Enqueue `name` on torch's current stream of the device the tensors live on; raise on a non-zero code.
    (A kernel launch goes to the calling thread's CURRENT device: tensors of another device -- a process that
    drives several GPUs without torch.cuda.set_device -- are launched under that device's guard.)"""
    lib = load()
    conv = _convert(name, args)
    di = -1
    for a in args:
        if isinstance(a, torch.Tensor):
            di = a.get_device()
            break
    if di >= 0 and di != torch.cuda.current_device():
        with torch.cuda.device(di):
            rc = getattr(lib, name)(*conv, current_stream_handle(di))
    else:
        rc = getattr(lib, name)(*conv, current_stream_handle())
    if rc != 0:
        raise DenetHipError(f"{name} failed: {strerror(rc)} (code {rc})")


def _convert(name, args):
    sig = _SIGS[name]
    if len(args) != len(sig):
        raise TypeError(f"{name}: expected {len(sig)} arguments, got {len(args)}")
    conv = []
    add = conv.append
    for i, (c, a) in enumerate(zip(sig, args)):
        if c in "pFL":
            # (fast path inline: ~50 arguments per fused-step call, twice per 90 us step)
            if a is None:
                add(None)
            elif isinstance(a, torch.Tensor):
                want = _WANT.get(c)
                if (want is not None and a.dtype != want) or not a.is_cuda or not a.is_contiguous():
                    _ptr(a, c, f"{name} argument {i}")          # raises with the reason
                add(a.data_ptr())
            else:
                add(int(a))
        elif c == "h":
            add(None if a is None else ctypes.cast(ctypes.pointer(a), ctypes.c_void_p))
        elif c == "f":
            add(float(a))
        else:
            add(int(a))
    return conv


def bind(name: str, *args):
    """Pre-convert the arguments of `name` once and return `run(stream_handle)`.  For loops that call
    the same entry point on the same buffers every iteration (the argument conversion of the ~40
    arguments of the fused step costs more host time than the launches themselves).  The caller keeps
    the tensors alive and unchanged in place (same data_ptr) for as long as it uses `run`."""
    lib = load()
    fn = getattr(lib, name)
    conv = tuple(_convert(name, args))
    keep = args                                            # the bound buffers stay referenced

    def run(stream):
        rc = fn(*conv, stream)
        if rc != 0:
            raise DenetHipError(f"{name} failed: {strerror(rc)} (code {rc})")
    run.keep = keep
    return run


def query(name: str, *args) -> int:
    return int(getattr(load(), name)(*[int(a) for a in args]))


def feat_code(t: torch.Tensor, *same) -> int:
    """DENET_F32 / DENET_BF16 for feature tensor `t`; every tensor in `same` (the other feature
    tensors of the call: one code describes them all) must have the same dtype."""
    for o in same:
        if o is not None and o.dtype != t.dtype:
            raise DenetHipError(f"feature tensors of one call must share a dtype, got {t.dtype} and {o.dtype}")
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise DenetHipError(f"features must be float32 or bfloat16, got {t.dtype}")
