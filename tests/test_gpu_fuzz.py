"""GPU: randomised parity sweep of the fused head step against the oracle (scripts/fuzz_parity.py):
random small shapes with class counts on both sides of every kernel threshold (8, 32, 64, 96, 128),
power-of-two and other up-sampling scales, 1-3 ways, 1-8 shots, C = 64 / 256.  Bar: 1e-3 relative
(BASELINE.json north_star); measured worst case over the sweep is ~2e-6."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("seed", [1, 2])
def test_random_shapes_match_oracle(seed):
    from fuzz_parity import sweep
    worst, failures = sweep(seed=seed, count=16, verbose=False)
    assert not failures, failures
    assert max(worst.values()) < 2e-5, worst
