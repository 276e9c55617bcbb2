"""The native host RNG (csrc/np_random.hip) must consume numpy's global MT19937 stream exactly like
the reference's per-row np.random.choice loops: same outputs, same generator state afterwards."""
import numpy as np


def _reference_groups(counts, size):
    out = np.full((len(counts), size), -1, dtype=np.int64)
    for r, n in enumerate(counts):
        if n >= size:
            out[r] = np.random.choice(n, size, replace=False)     # get_regiondataset.py:333-335
        elif n > 0:
            out[r] = np.random.choice(n, size, replace=True)      # get_regiondataset.py:336-337
    return out


def _reference_crops(counts, size):
    out = np.zeros((len(counts), size), dtype=np.int64)
    valid = np.zeros(len(counts), bool)
    for r, n in enumerate(counts):
        length = n
        if n > size:
            out[r] = np.random.choice(n, size, replace=False)     # gripper_region_network.py:533-535
            length = size
        elif n > 5:
            out[r] = np.random.choice(n, size, replace=True)      # gripper_region_network.py:536-537
            length = size
        valid[r] = length > 5                                     # :538 tests the RESAMPLED index
    return out, valid


def test_choice_rows_matches_numpy_stream():
    from regnet_for_3d_grasping_amd.np_random import choice_rows
    rng = np.random.default_rng(0)
    for seed in (0, 1, 123, 2 ** 31 - 1, 987654321):
        for size in (1, 64, 256, 1024):
            counts = np.concatenate([rng.integers(0, 3 * size + 5, 40), [0, 1, 2, 5, 6, size - 1, size, size + 1,
                                                                        2 ** 16, 2 ** 16 + 1, 40000]]).astype(np.int64)
            np.random.seed(seed)
            np.random.random(seed % 700)                    # start from an arbitrary position in the block
            want = _reference_groups(counts, size)
            after = np.random.randint(0, 2 ** 31 - 1, 5)
            np.random.seed(seed)
            np.random.random(seed % 700)
            got, valid = choice_rows(counts, size, 0)
            assert np.array_equal(got, want)
            assert np.array_equal(valid, counts > 0)
            assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, 5), after)   # generator left in the same state

            np.random.seed(seed + 1)
            want, wvalid = _reference_crops(counts, size)
            after = np.random.randint(0, 2 ** 31 - 1, 5)
            np.random.seed(seed + 1)
            got, valid = choice_rows(counts, size, 1)
            assert np.array_equal(got, want) and np.array_equal(valid, wvalid)
            assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, 5), after)


def test_choice_rows_shapes_and_gaussian_cache_preserved():
    from regnet_for_3d_grasping_amd.np_random import choice_rows
    np.random.seed(5)
    np.random.standard_normal()                 # leaves a cached gaussian in the legacy state
    state = np.random.get_state()
    pos, valid = choice_rows(np.array([[3, 0], [700, 64]]), 64, 0)
    assert pos.shape == (2, 2, 64) and valid.tolist() == [[True, False], [True, True]]
    assert (pos[0, 1] == -1).all() and sorted(pos[1, 1].tolist()) == list(range(64))
    assert np.random.get_state()[3] == state[3] and np.random.get_state()[4] == state[4]
