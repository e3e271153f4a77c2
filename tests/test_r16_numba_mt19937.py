"""R16 pin: the reference draws its initial positions from numba's CPU
Mersenne Twister, seeded inside a jitted function (disimpy/simulations.py:346-350) and read
through ``np.random.random(k)`` (:359, :375, :392).  ``disimpy_amd.simulations._host_rng``
restates that stream with ``numpy.random.RandomState(seed)``.

numba itself cannot be imported here, but its source can be READ (numba 0.54.1 under
/opt/conda/lib/python3.9/site-packages/numba; the reference pins >= 0.58.1, the files below have
not changed in substance since 0.2x).  This test is a pure-Python restatement of exactly those
lines, compared with RandomState for seeds {0, 1, 123}:

* ``np.random.seed(uint32)`` lowers to ``numba_rnd_init(state, seed)``
  (numba/cpython/randomimpl.py:207-219), which is MT19937 ``init_genrand``:
  ``mt[pos] = seed; seed = 1812433253 * (seed ^ seed >> 30) + pos + 1`` and ``index = 624``
  (numba/_random.c:58-71);
* ``np.random.random`` lowers to ``get_next_double`` (randomimpl.py:227-234, :133-146):
  ``a = next32() >> 5; b = next32() >> 6; (a * 67108864.0 + b) / 9007199254740992.0``;
* ``get_next_int32`` (randomimpl.py:108-131): reshuffle when ``index >= 624``
  (``numba_rnd_shuffle``, _random.c:36-55: the standard MT19937 twist with M = 397 and
  0x9908b0df), then the standard tempering (11, 7 & 0x9d2c5680, 15 & 0xefc60000, 18);
* ``np.random.random(k)`` fills its k elements in index order from that stream.

With this the row R16 is "read from source" rather than "assumed".
"""
import numpy as np
import pytest

N, M_ = 624, 397
U32 = 0xFFFFFFFF


class NumbaMT:
    def __init__(self, seed):  # numba_rnd_init, _random.c:58-71
        seed &= U32
        self.mt = []
        for pos in range(N):
            self.mt.append(seed)
            seed = (1812433253 * (seed ^ (seed >> 30)) + pos + 1) & U32
        self.index = N

    def shuffle(self):  # numba_rnd_shuffle, _random.c:36-55
        mt = self.mt
        for i in range(N - M_):
            y = (mt[i] & 0x80000000) | (mt[i + 1] & 0x7FFFFFFF)
            mt[i] = mt[i + M_] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        for i in range(N - M_, N - 1):
            y = (mt[i] & 0x80000000) | (mt[i + 1] & 0x7FFFFFFF)
            mt[i] = mt[i + (M_ - N)] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        y = (mt[N - 1] & 0x80000000) | (mt[0] & 0x7FFFFFFF)
        mt[N - 1] = mt[M_ - 1] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)

    def next32(self):  # get_next_int32, randomimpl.py:108-131
        if self.index >= N:
            self.shuffle()
            self.index = 0
        y = self.mt[self.index]
        self.index += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & U32

    def random(self):  # get_next_double, randomimpl.py:133-146
        a = self.next32() >> 5
        b = self.next32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0


@pytest.mark.parametrize("seed", [0, 1, 123])
def test_numba_cpu_stream_is_numpy_legacy_randomstate(seed):
    mt = NumbaMT(seed)
    want = np.array([mt.random() for _ in range(2000)])  # crosses three reshuffles (2 x 2000 / 624)
    got = np.random.RandomState(seed).random_sample(2000)
    assert np.array_equal(got, want)


def test_fill_sphere_follows_the_reference_loop_on_the_numba_stream():
    """disimpy/simulations.py:369-382 run on the restated numba stream == the batched host sampler."""
    from disimpy_amd import simulations

    seed, n, radius = 123, 200, 5e-6
    mt = NumbaMT(seed)
    pts = []
    while len(pts) < n:
        p = (np.array([mt.random(), mt.random(), mt.random()]) - 0.5) * 2 * radius
        if np.linalg.norm(p) < radius:
            pts.append(p)
    got = simulations._fill_sphere(n, radius, simulations._host_rng(seed))
    assert np.array_equal(got, np.array(pts))
    # a shard draws only the prefix it needs: rows [a, b) of the same stream
    pos, _ = simulations._initial_positions(n, __import__("disimpy_amd").substrates.sphere(radius), seed, True, 0, (50, 120))
    assert np.array_equal(pos, np.array(pts)[50:120])
