import numpy as np


def test_numpy_uniform_generator_matches_c(oracle_mod):
    from deltaq_amd import workload as wl
    from tools import datagen
    for n in (0, 1, 7, 8, 9, 1000, 123457):
        assert np.array_equal(wl.gen_uniform(n, 0x5EED0002), datagen.gen_uniform(n, 0x5EED0002))


def test_enwik_like_is_deterministic_and_skewed():
    from tools import datagen
    a = datagen.gen_enwik_like(200_000, 0xD17A0)
    b = datagen.gen_enwik_like(200_000, 0xD17A0)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:100_000], datagen.gen_enwik_like(100_000, 0xD17A0))   # prefix-stable
    hist = np.bincount(a, minlength=256)
    assert (hist > 0).sum() < 100 and hist[ord(" ")] > 0.1 * a.size
