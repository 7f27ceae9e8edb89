"""CPU: the C-ABI library loads and exports every symbol include/denet_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "denet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(denet_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from denet_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/denet_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_version_and_strerror():
    from denet_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.denet_version()
    assert _lib.strerror(0) == "ok"
    assert "workspace" in _lib.strerror(-4)


def test_workspace_queries_are_pure_host_calls():
    from denet_amd import _lib
    assert _lib.query("denet_classify_bwd_parts", 1681) == 27
    assert _lib.query("denet_proto_extend_bwd_workspace", 8, 1, 5, 21, 256) == (3 * 40 * 256 + 8 * 21 * 256) * 4
    assert _lib.query("denet_head_step_workspace", 8, 1, 5, 21, 256, 41, 41, 321, 321) > 0
    assert _lib.query("denet_head_step_workspace", 0, 1, 5, 21, 256, 41, 41, 321, 321) < 0
    assert _lib.query("denet_upsample_loss_parts", 21, 41, 41, 321, 321, 1) % 41 == 0
    # up-sampling only, as the reference (loss/losses.py:29-37): label maps smaller than the logits are refused
    assert _lib.query("denet_upsample_loss_parts", 21, 41, 41, 40, 321, 1) < 0
    assert _lib.query("denet_upsample_loss_parts", 21, 41, 41, 321, 40, 0) < 0
    assert _lib.query("denet_head_step_workspace", 8, 1, 5, 21, 256, 41, 41, 33, 33) < 0
    assert _lib.query("denet_upsample_loss_parts", 21, 41, 41, 41, 41, 1) > 0          # same size is fine


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from denet_amd import ops, _lib
    with pytest.raises(_lib.DenetHipError):
        ops.mask_resize(torch.zeros(1, 4, 4, dtype=torch.int64), 2, 2)


def test_wrong_dtype_is_rejected_before_any_launch():
    """The kernels read raw memory: a bf16 / half / int32 tensor where float32 / int64 is expected, or
    Fq and Fs of different dtypes under one dtype code, must raise instead of being misread."""
    import torch
    from denet_amd import _lib
    with pytest.raises(_lib.DenetHipError, match="float32"):
        _lib._convert("denet_batch_mean", (torch.zeros(2, 4, dtype=torch.bfloat16), 2, 4, torch.zeros(4)))
    with pytest.raises(_lib.DenetHipError, match="int64"):
        _lib._convert("denet_mask_resize", (torch.zeros(1, 4, 4, dtype=torch.int32), 1, 4, 4, 2, 2,
                                            torch.zeros(1, 4), torch.zeros(1)))
    with pytest.raises(_lib.DenetHipError, match="share a dtype"):
        _lib.feat_code(torch.zeros(1), torch.zeros(1, dtype=torch.bfloat16))
    assert _lib.feat_code(torch.zeros(1, dtype=torch.bfloat16), torch.zeros(2, dtype=torch.bfloat16)) == _lib.BF16
