"""A fixed set of seeds of the randomised parity sweep (tests/fuzz_parity.py): random scene shapes, topologies, blends,
filters, viewports, formats, sample counts and load ops, HIP against the oracle bit for bit."""
import pytest

import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first_seed", [0, 1000, 2000])
def test_random_scenes_match_the_oracle(oracle, first_seed):
    for seed in range(first_seed, first_seed + 25):
        err, what = fuzz_parity.run_one(seed)
        assert err is None, f"seed {seed}: {err} {what}"


def test_random_lz4_streams_match_liblz4(oracle):
    import fuzz_codec
    for seed in range(5):
        err = fuzz_codec.run_round(seed)
        assert err is None, f"seed {seed}: {err}"
