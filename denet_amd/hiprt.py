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
        _hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        _hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    return _hip


# hipEventDisableSystemFence (hip_runtime_api.h): no system-scope fence when the event completes.  For events that
# only measure time: a default event writes back and invalidates the caches when it is reached, which the
# bracketed kernel and the one after it then pay for (the probe read 20.1 us for a kernel rocprofv3 times at 17.2,
# and a probed step was 14 us longer than an unprobed one).
EVENT_DISABLE_SYSTEM_FENCE = 0x20000000


class Event:
    """Timing-only event (hipEventElapsedTime between two of them); not for host-visible synchronisation of data."""

    def __init__(self):
        self.h = ctypes.c_void_p()
        rc = hip().hipEventCreateWithFlags(ctypes.byref(self.h), EVENT_DISABLE_SYSTEM_FENCE)
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags failed ({rc})")

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
