"""Host-side logic that needs neither a GPU nor the oracle."""
import numpy as np
import pytest


def test_shard_range_partitions_exactly():
    from gym_reinmav_amd.distributed import shard_range

    for n, w in [(1048576, 8), (65536, 1), (10, 4), (7, 8), (0, 3)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for s, c in spans:
            assert s == pos
            pos += c
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(1048576, 3, 8) == (393216, 131072)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def test_registry_ids_match_reference():
    import gym_reinmav_amd as g

    assert sorted(g.ENTRY_POINTS) == ["quadrotor2d-slungload-v0", "quadrotor2d-v0", "quadrotor3d-slungload-v0",
                                      "quadrotor3d-v0"]
    assert g.ENV_IDS["quadrotor3d-v0"] == "quad3d"
    with pytest.raises(KeyError):
        g.make("reinmav-v0")  # out of scope, not silently mapped to something else


def test_box_space():
    from gym_reinmav_amd.spaces import Box

    b = Box(low=0.0, high=10.0, shape=(4,), dtype=np.float32)
    assert b.shape == (4,) and b.contains(b.sample()) and not b.contains(np.full(4, 11.0, np.float32))
