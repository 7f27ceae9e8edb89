"""Allocation-free driver of the fused head step (include/denet_hip.h, T1).

All device buffers (workspace, logits, losses, gradients) are allocated once;
`forward()` / `backward()` only enqueue kernels on torch's current stream.  The
seven parameter gradients live in ONE flat fp32 bucket so that data-parallel
training needs a single RCCL all-reduce per step.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import call, feat_code, query

PARAM_ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
               "map.linear.weight", "map.linear.bias"]


class HeadStepEngine:
    def __init__(self, N, way, shot, K, C, h, w, H, W, device, feat_dtype=torch.float32):
        self.dims = (N, way, shot, K, C, h, w, H, W)
        self.device = device
        self.feat_dtype = feat_dtype
        f32 = dict(dtype=torch.float32, device=device)
        self.ws_bytes = query("denet_head_step_workspace", N, way, shot, K, C, h, w, H, W)
        if self.ws_bytes < 0:
            raise _lib.DenetHipError(_lib.strerror(self.ws_bytes))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.logits_full = torch.empty(N, K, h, w, **f32)
        self.logits_binary = torch.empty(N * way, 2, h, w, **f32)
        self.losses = torch.zeros(2, **f32)
        self.dFq = torch.empty(N, C, h, w, dtype=feat_dtype, device=device)
        self.dFs = torch.empty(N, way, shot, C, h, w, dtype=feat_dtype, device=device)
        sizes = [K * C, C * C, C, C * C, C, C * C, C]
        self.grad_bucket = torch.zeros(sum(sizes), **f32)
        self.grads = {}
        off = 0
        shapes = [(K, C), (C, C, 1, 1), (C,), (C, C, 1, 1), (C,), (C, C, 1, 1), (C,)]
        for name, n, shp in zip(PARAM_ORDER, sizes, shapes):
            self.grads[name] = self.grad_bucket[off:off + n].view(shp)
            off += n
        # one probe record for the engine's lifetime: the library reads it at every call (host pointer), so
        # switching the probe on / off or to another event pair is a field write, not a re-bind
        self._probe = _lib.Probe()
        self._probe.which = 0
        self._deferred = False
        self._have_focal = True
        self._bound = None

    def set_probe(self, which, start_event, stop_event):
        """Bracket kernel `which` (a name in _lib.KERNEL_IDS, or None) with HIP events."""
        p = self._probe
        if which is None:
            p.which = 0
            p.start = None
            p.stop = None
            return
        p.start = start_event.h
        p.stop = stop_event.h
        p.which = _lib.KERNEL_IDS[which]
        self._probe_events = (start_event, stop_event)         # keep them alive while the record names them

    # ---- bound step: the same buffers every iteration (training loops over resident batches, bench) ----
    def bind_step(self, Fq, Fs, Ys, labels, Yq, Yq_full, wgt, params, gate_mask=None, weight_mask=None,
                  grad_loss_k=None, drop=None):
        """Fix the step's arguments (with_grad, deferred losses): run_forward() / run_backward(phase)
        then only make the library calls -- the argument conversion is done once here.  The tensors must
        stay alive and in place.  drop: a _lib.Dropout (train mode, masks generated in the kernels); the
        library reads it at every call, so advance `drop.seed` in place between steps."""
        self._bound = self.make_bound(Fq, Fs, Ys, labels, Yq, Yq_full, wgt, params, gate_mask, weight_mask,
                                      grad_loss_k, drop)

    def use_bound(self, bound):
        """Switch to another pre-bound argument set (make_bound): a loop that rotates over several resident
        batches binds each once and switches between them with this field write."""
        self._bound = bound

    def make_bound(self, Fq, Fs, Ys, labels, Yq, Yq_full, wgt, params, gate_mask=None, weight_mask=None,
                   grad_loss_k=None, drop=None):
        """The pre-converted calls of one step on these input tensors (outputs, gradients and workspace are the
        engine's own): what bind_step installs."""
        N, way, shot, K, C, h, w, H, W = self.dims
        # the pre-bound calls launch on the calling thread's current device with no per-step check (they are the
        # 90 us loop): refuse to bind while another device is current
        dev_index = torch.device(self.device).index
        if dev_index is not None and dev_index != torch.cuda.current_device():
            raise _lib.DenetHipError(f"HeadStepEngine on cuda:{dev_index} bound while cuda:{torch.cuda.current_device()} "
                                     "is the current device: torch.cuda.set_device first (one process per GPU)")
        P = [params[k] for k in PARAM_ORDER]
        deferred = Yq_full is not None
        have_focal = Yq is not None
        g = self.grads
        fwd = _lib.bind("denet_head_step_train_fwd", Fq, Fs, feat_code(Fq, Fs), Ys, labels, Yq, Yq_full, wgt, *P,
                        gate_mask, weight_mask, drop, N, way, shot, K, C, h, w, H, W,
                        _lib.GRAD_DEFER_LOSSES if deferred else 1, self.logits_full, self.logits_binary,
                        self.losses, self.ws, self.ws_bytes, self._probe)
        bwd = {}
        for phase, fl in ((0, 0), (1, _lib.BWD_NO_DFS), (2, _lib.BWD_ONLY_DFS)):
            bwd[phase] = _lib.bind(
                "denet_head_step_train_bwd", Fq, feat_code(Fq, self.dFq, self.dFs), labels,
                params["gam.gate.0.weight"], params["gam.gate.2.weight"], params["map.linear.weight"], gate_mask,
                weight_mask, drop, grad_loss_k,
                N, way, shot, K, C, h, w, H, W, self.dFq, self.dFs, g["weight"], g["gam.gate.0.weight"],
                g["gam.gate.0.bias"], g["gam.gate.2.weight"], g["gam.gate.2.bias"], g["map.linear.weight"],
                g["map.linear.bias"], self.losses if deferred else None,
                (_lib.BWD_HAVE_FOCAL if have_focal else 0) | fl, self.ws, self.ws_bytes, self._probe)
        return (fwd, bwd, deferred, have_focal)

    def run_forward(self, stream=None):
        fwd, _, self._deferred, self._have_focal = self._bound
        fwd(torch.cuda.current_stream().cuda_stream if stream is None else stream)

    def run_backward(self, phase=0, stream=None):
        self._bound[1][phase](torch.cuda.current_stream().cuda_stream if stream is None else stream)

    def forward(self, Fq, Fs, Ys, labels, Yq, Yq_full, wgt, params, gate_mask=None, weight_mask=None,
                with_grad=True, defer_losses=False, drop=None):
        """drop: a _lib.Dropout for train mode (pass the same one to backward()).  defer_losses: the loss scalars are reduced inside the first backward launch instead of a
        launch of their own (self.losses is then valid after backward(); needs with_grad and Yq_full)."""
        N, way, shot, K, C, h, w, H, W = self.dims
        P = [params[k] for k in PARAM_ORDER]
        self._deferred = bool(defer_losses and with_grad and Yq_full is not None)
        self._have_focal = Yq is not None
        mode = _lib.GRAD_DEFER_LOSSES if self._deferred else (1 if with_grad else 0)
        call("denet_head_step_train_fwd", Fq, Fs, feat_code(Fq, Fs), Ys, labels, Yq, Yq_full, wgt, *P, gate_mask,
             weight_mask, drop, N, way, shot, K, C, h, w, H, W, mode, self.logits_full, self.logits_binary,
             self.losses, self.ws, self.ws_bytes, self._probe)

    def backward(self, Fq, labels, params, gate_mask=None, weight_mask=None, grad_loss_k=None, phase=0, drop=None):
        """phase 0: the whole backward.  phase 1: every gradient except dFs (the parameter-gradient
        bucket is complete afterwards); phase 2: dFs.  Data-parallel callers start the bucket's
        all-reduce between phases 1 and 2 so that it overlaps with the dFs stream."""
        N, way, shot, K, C, h, w, H, W = self.dims
        g = self.grads
        flags = (_lib.BWD_HAVE_FOCAL if self._have_focal else 0) | {0: 0, 1: _lib.BWD_NO_DFS, 2: _lib.BWD_ONLY_DFS}[phase]
        call("denet_head_step_train_bwd", Fq, feat_code(Fq, self.dFq, self.dFs), labels, params["gam.gate.0.weight"],
             params["gam.gate.2.weight"], params["map.linear.weight"], gate_mask, weight_mask, drop, grad_loss_k,
             N, way, shot, K, C, h, w, H, W, self.dFq, self.dFs, g["weight"], g["gam.gate.0.weight"],
             g["gam.gate.0.bias"], g["gam.gate.2.weight"], g["gam.gate.2.bias"], g["map.linear.weight"],
             g["map.linear.bias"], self.losses if self._deferred else None, flags,
             self.ws, self.ws_bytes, self._probe)
