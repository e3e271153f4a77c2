"""The periodic image of a subvoxel coordinate as the cooperative mesh kernel works it out since round 6
(csrc/dsim_mesh_coop.h, wrap_subvoxel): sn = floor((float(s) + 0.5f) * (1.0f / n)) instead of the integer floor division of
disimpy/simulations.py:661, :675 (math.floor(x / n)).  Checked here, in the same float32 operations, for EVERY grid size the
library admits (n <= 4096, upload_mesh) and every coordinate the traversal can hand in (|s| <= 33 n: anything beyond +-32
voxels goes to the serial pass, plan_runs)."""
import numpy as np


def _wrap_f32(s, n):
    inv_n = np.float32(1.0) / np.float32(n)  # the host's MeshDev::inv_n*
    t = (s.astype(np.float32) + np.float32(0.5)) * inv_n  # two roundings, as on the device (-ffp-contract=off)
    return np.floor(t).astype(np.int64)


def test_float32_image_number_is_the_exact_floor_division_for_every_admitted_grid():
    worst = 1.0
    for n in range(1, 4097):
        s = np.arange(-33 * n, 33 * n + 1, dtype=np.int64)
        got = _wrap_f32(s, n)
        want = s // n  # Python floor division
        assert np.array_equal(got, want), (n, s[got != want][:5])
        if n in (1, 50, 4096):  # how far the product stays from an integer boundary: never closer than ~1 / (2 n) minus rounding
            t = (s.astype(np.float64) + 0.5) / n
            worst = min(worst, float(np.min(np.abs(t - np.round(t)))) * n)
    assert worst > 0.49


def test_wrapped_coordinate_lands_in_the_grid():
    rng = np.random.RandomState(3)
    for n in (1, 2, 7, 50, 1023, 4096):
        s = rng.randint(-33 * n, 33 * n + 1, size=100000).astype(np.int64)
        sn = _wrap_f32(s, n)
        x = s - sn * n
        assert x.min() >= 0 and x.max() <= n - 1
