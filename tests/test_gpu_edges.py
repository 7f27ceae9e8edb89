"""GPU parity on the edges of the shape space, as sweeps kept in scripts/diag (each compares the fused step with
the oracle and exits non-zero on a mismatch): the smallest feature maps (4-12 positions, both sides of the
backward's 8-position fragments), 4-12 ways with duplicate labels (more ways than a block has waves), and
degenerate label maps (binary labels all ignored, an empty support mask, a one-class query map, one-hot class
weights), and up-sampling factors of 16x-64x (each case matches or is refused with DenetHipError)."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu

DIAG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "diag")


@pytest.mark.parametrize("script", ["tiny_shapes.py", "many_ways.py", "degenerate_labels.py", "big_scales.py"])
def test_edge_sweep(script, capsys):
    with pytest.raises(SystemExit) as ex:
        runpy.run_path(os.path.join(DIAG, script), run_name="__main__")
    out = capsys.readouterr().out
    assert ex.value.code == 0, out[-2000:]
    assert "failures: 0" in out
