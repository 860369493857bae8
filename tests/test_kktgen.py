"""CPU: the random stream of the synthetic bench workloads (xoshiro256** seeded by splitmix64(20260923)) -- known answers of the two
published algorithms, and the generator's by-construction inertia on a small sibling of `synth_1e6` checked with the oracle."""
import numpy as np

from oracle import kkt_oracle as ko
from tests.support import kktgen


def test_splitmix64_and_xoshiro256starstar_known_answers():
    x = kktgen.Xoshiro256(0)
    assert int(x.state[0]) == 0xE220A8397B1DCDAF and int(x.state[1]) == 0x6E789E6AA1B965F4      # splitmix64 from seed 0: its first two outputs
    x.state[:] = [1, 2, 3, 4]
    # xoshiro256** from the state (1, 2, 3, 4): 11520, 0, 1509978240, 1215971899390074240 -> the top 53 bits of each
    assert np.array_equal(x.random(4) * 2.0 ** 53, [11520 >> 11, 0, 1509978240 >> 11, 1215971899390074240 >> 11])
    a = kktgen.Xoshiro256(20260923).uniform(-1, 1, (3, 5)); b = kktgen.Xoshiro256(20260923).uniform(-1, 1, 15)
    assert np.array_equal(a.ravel(), b) and np.all(np.abs(a) < 1)                                 # one stream, consumed in C order


def test_synthetic_generator_on_the_xoshiro_stream_has_the_inertia_it_was_built_for():
    n, r, c, v, neg = kktgen.grid_kkt(12, 10, dof=3, ncon=2, seed=20260923, sigma_exp=8.0, rng="xoshiro")
    n2, r2, c2, v2, _ = kktgen.grid_kkt(12, 10, dof=3, ncon=2, seed=20260923, sigma_exp=8.0, rng="xoshiro")
    assert np.array_equal(v, v2) and np.array_equal(r, r2) and neg == 12 * 10 * 2               # deterministic
    K = kktgen.to_scipy(n, r, c, v)
    b = K @ np.ones(n)
    x, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=0.01)
    assert oneg == neg and ozero == 0 and np.abs(x - 1).max() <= 1e-6
