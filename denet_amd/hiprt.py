"""Minimal ctypes view of the HIP runtime: events for timing a kernel on the
stream it is launched on (torch.cuda.Event only sees torch's current stream and
cannot be handed to the C ABI)."""
from __future__ import annotations

import ctypes

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        _hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    return _hip


class Event:
    def __init__(self):
        self.h = ctypes.c_void_p()
        rc = hip().hipEventCreate(ctypes.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"hipEventCreate failed ({rc})")

    def record(self, stream):
        rc = hip().hipEventRecord(self.h, ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed ({rc})")

    def synchronize(self):
        hip().hipEventSynchronize(self.h)

    def elapsed_ms(self, stop: "Event") -> float:
        out = ctypes.c_float()
        rc = hip().hipEventElapsedTime(ctypes.byref(out), self.h, stop.h)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(out.value)

    def __del__(self):
        try:
            if self.h:
                hip().hipEventDestroy(self.h)
        except Exception:
            pass
