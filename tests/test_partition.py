"""The row-tile partition of a frame over the shards of a multi-GPU render (rm_internal.h tile_owner / tile_of / shard_rows,
exported as rm_shard_rows / rm_shard_row_to_frame), with and without root relief (rm_set_root_relief): every frame row belongs
to exactly one shard, a shard's rows are packed in frame order, the row counts add up, rank 0 is relieved by the stated share —
checked by brute force over many (H, tileRows, N, K).  No GPU needed (host functions of the library)."""
import itertools

import pytest

from raymarcher_amd import lib


@pytest.fixture(autouse=True)
def _restore_relief():
    yield
    lib().rm_set_root_relief(0)


def test_partition_is_exact_for_every_relief():
    L = lib()
    cases = list(itertools.product([1, 7, 8, 9, 50, 64, 270, 2160, 2161], [1, 3, 8], [1, 2, 3, 4, 8], [0, 2, 3, 8, 16]))
    cases += [(4320, 8, 8, 8), (4320, 8, 4, 8), (1080, 8, 8, 5), (17, 8, 8, 8), (3, 8, 8, 2)]
    for H, T, N, K in cases:
        assert L.rm_set_root_relief(K) == 0 and L.rm_get_root_relief() == K
        seen = {}
        rows = [L.rm_shard_rows(H, T, k, N) for k in range(N)]
        assert sum(rows) == H, (H, T, N, K, rows)
        for k in range(N):
            prev = -1
            for r in range(rows[k]):
                y = L.rm_shard_row_to_frame(H, T, k, N, r)
                assert 0 <= y < H and y not in seen, (H, T, N, K, k, r, y)
                assert y > prev  # a shard's rows are packed in frame order
                # rows of one tile stay together: local rows r and r+1 are neighbours unless a tile ends between them
                if r % T:
                    assert y == prev + 1
                seen[y] = k
                prev = y
            assert L.rm_shard_row_to_frame(H, T, k, N, rows[k]) == -1
        assert len(seen) == H
        assert max(rows) == max(rows[:2])  # the largest shard is shard 0 or 1: what gather slots are sized by
        if K == 0:
            for y, k in seen.items():
                assert k == (y // T) % N
    assert L.rm_set_root_relief(1) != 0 and L.rm_set_root_relief(65) != 0 and L.rm_set_root_relief(-3) != 0


def test_root_relief_gives_rank_0_its_stated_share():
    L = lib()
    H, T = 2160, 8  # 270 tiles
    for N, K in ((8, 8), (4, 8), (2, 16), (8, 4)):
        L.rm_set_root_relief(K)
        rows = [L.rm_shard_rows(H, T, k, N) for k in range(N)]
        peers = sum(rows[1:]) / (N - 1)
        assert abs(rows[0] / peers - (K - 1) / K) < 0.08, (N, K, rows)
        assert max(rows[1:]) - min(rows[1:]) <= T  # the peers stay balanced to within one tile
    L.rm_set_root_relief(0)
    rows = [L.rm_shard_rows(H, T, k, 8) for k in range(8)]
    assert rows[0] == max(rows) and max(rows) - min(rows) <= T
