"""CPU: the measurement helpers of bench.py that need no GPU — how many buffer sets a call needs so that a rotation over them
finds its bytes in HBM and not in the 256 MiB Infinity Cache (cold_sets), and how many cores the all-cores CPU baseline may use."""
import os

import bench


def test_cold_sets_keep_512_mib_of_other_traffic_between_two_uses():
    for footprint in (1, 12 * 10_000_000, 21 * 10_000_000, 152 * 1_000_000, 26 * 10_000_000, 400 << 20, 512 << 20, 2 << 30):
        k = bench.cold_sets(footprint)
        assert k >= 2                                               # never a replay on one set
        assert (k - 1) * footprint >= bench.COLD_TRAFFIC            # the others' bytes between two uses of a set
        assert k == 2 or (k - 2) * footprint < bench.COLD_TRAFFIC   # ... and no set more than that needs
    assert bench.COLD_TRAFFIC >= 2 * (256 << 20)                    # twice the Infinity Cache
    assert bench.cold_sets(120_000_000) == 6 and bench.cold_sets(210_000_000) == 4      # cfg 3 / the dual-index sheet at 10 M rows


def test_usable_cores_is_what_the_process_may_use():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n <= len(os.sched_getaffinity(0))
