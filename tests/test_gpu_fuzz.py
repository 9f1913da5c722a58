"""Seeded fuzz of the fused substep against the C oracle: random grid shapes (4 .. 128 columns, 4 .. 64 rows, 3 .. 40 levels, odd and
non-power-of-two ones among them), closures, scalars, stretched levels, floor, the three lids, one-GPU / slab layout with 1 / 2 / 4
transpose chunks, and the order switches (UDC_PTOTAL, UDC_P_TRANSPOSE, UDC_MOM_PIPE), three or six substeps each, 1e-9.
tests/fuzz_parity.py is the generator (1150 cases of six seeds ran clean in round 6: profiles/r06/fuzz_parity.txt); here 60
cases of a fixed seed."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sixty_random_configurations_match_the_oracle():
    import fuzz_parity
    rng = np.random.default_rng(20260930)
    bad = [i for i in range(60) if fuzz_parity.one(rng, i)]
    assert not bad, bad


def test_twenty_random_decompositions_match_one_rank():
    """tests/fuzz_slabs.py: random grids on 2 / 4 / 8 virtual ranks against one rank (360 cases ran clean in round 6); 20 of a fixed seed."""
    import fuzz_slabs
    rng = np.random.default_rng(20260930)
    bad = [i for i in range(20) if fuzz_slabs.one(rng, i)]
    assert not bad, bad

