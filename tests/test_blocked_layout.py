"""The tile-blocked batch layout on the CPU: sk_blocked_layout_init's arithmetic, and the two packers (numpy in
seqkit_amd/capi.py for the tests, torch in bench.py for the device-resident shard) put every byte where the header says."""
import numpy as np
import pytest

from seqkit_amd import capi


@pytest.mark.parametrize("n_mates,stride,bc_stride,flags", [(2, 150, 17, 3), (1, 150, 0, 3), (2, 101, 8, 3 | 4 | 8), (1, 33, 32, 2), (2, 960, 17, 1)])
def test_layout_arithmetic(hip_lib, n_mates, stride, bc_stride, flags):
    lay = capi.blocked_layout(n_mates, stride, bc_stride, flags)
    segs = []
    for m in range(n_mates):
        segs.append((lay.in_qual[m], 64 * stride))
        if flags & capi.SK_BLK_MASK:
            segs.append((lay.in_seq[m], 64 * stride))
        else:
            assert lay.in_seq[m] == -1
        if flags & capi.SK_BLK_LEN:
            segs.append((lay.in_len[m], 128))
        else:
            assert lay.in_len[m] == -1
    if bc_stride:
        segs.append((lay.in_bc, 64 * bc_stride))
    else:
        assert lay.in_bc == -1 and lay.out_assign == -1
    segs.sort()
    assert segs[0][0] == 0 and all(o % 64 == 0 for o, _ in segs)
    assert all(a + la <= b for (a, la), (b, _) in zip(segs, segs[1:]))            # segments do not overlap
    assert segs[-1][0] + segs[-1][1] <= lay.in_block and lay.in_block % 128 == 0 and lay.out_block % 128 == 0
    outs = []
    for m in range(n_mates):
        if flags & capi.SK_BLK_MASK:
            outs.append((lay.out_seq[m], 64 * stride))
        if flags & capi.SK_BLK_TRIM:
            outs.append((lay.out_lowest_k[m], 128))
    if bc_stride:
        outs.append((lay.out_assign, 256))
        if flags & capi.SK_BLK_DETAIL:
            outs += [(lay.out_lowest_diff, 64), (lay.out_first_idx, 128), (lay.out_last_idx, 128)]
    outs.sort()
    assert all(a + la <= b for (a, la), (b, _) in zip(outs, outs[1:])) and outs[-1][0] + outs[-1][1] <= lay.out_block


def test_layout_rejects_bad_shapes(hip_lib):
    from seqkit_amd import SeqkitHipError
    for args in ((0, 150, 17, 3), (3, 150, 17, 3), (2, 0, 17, 3), (2, 150, 17, 0), (2, 150, -1, 3), (2, 150, 17, 64), (2, 70000, 17, 3)):
        with pytest.raises(SeqkitHipError):
            capi.blocked_layout(*args)


def test_numpy_and_torch_packers_agree(hip_lib):
    import torch
    import bench
    lay = capi.blocked_layout(2, 150, 17, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
    n = 64 * 5 + 11
    nt = (n + 63) // 64
    rng = np.random.default_rng(3)
    seq = [rng.integers(1, 255, (nt * 64, 150), dtype=np.uint8) for _ in range(2)]
    qual = [rng.integers(1, 255, (nt * 64, 150), dtype=np.uint8) for _ in range(2)]
    bc = rng.integers(1, 255, (nt * 64, 17), dtype=np.uint8)
    for a in seq + qual + [bc]:
        a[n:] = 0                                                              # the numpy packer leaves the padding rows zero
    a = lay.pack([(seq[i][:n], qual[i][:n], None) for i in range(2)], bc[:n])
    b, bout = bench.pack_blocked(torch, lay, [torch.from_numpy(x) for x in seq], [torch.from_numpy(x) for x in qual], torch.from_numpy(bc), nt)
    assert a.size == lay.in_bytes(n) and np.array_equal(a, b.numpy()) and bout.numel() == lay.out_bytes(n)
    r = 64 * 4 + 7                                                             # a row of the last, partial tile
    t, rr = divmod(r, 64)
    assert np.array_equal(a[t * lay.in_block + lay.in_seq[1] + rr * 150:][:150], seq[1][r])
    assert np.array_equal(a[t * lay.in_block + lay.in_bc + rr * 17:][:17], bc[r])
    out = rng.integers(0, 255, lay.out_bytes(n), dtype=np.uint8)
    u = lay.unpack(out, n)
    v = bench.unpack_blocked(torch, lay, torch.from_numpy(out), nt)
    for i in range(2):
        assert np.array_equal(u["out_seq"][i], v["out_seq"][i].numpy()[:n])
        assert np.array_equal(u["lowest_k"][i], v["lowest_k"][i].numpy().view(np.uint16)[:n])
    assert np.array_equal(u["assign"], v["assign"].numpy()[:n])
