"""GPU: the count reduce of SURVEY.md §8(e) behind the C-ABI (RCCL inside libseqkit_hip.so).  The GPU box has one device,
so these cover what one device can: a communicator of one rank, and several ctxs of one process on the same device
(summed on the device; the RCCL leg between distinct devices is the same call with more leaders)."""
import numpy as np
import pytest

from seqkit_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_counts_allreduce_over_ctxs_sharing_a_device(hip_lib, oracle):
    import seqkit_amd
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    n = 30000
    bc, _ = synth.observe_barcodes(table, n, seed=77, halves=2)
    cuts = [0, 7001, 7001, 20000, n]                      # uneven shards, one of them empty
    ctxs = [seqkit_amd.Context(0) for _ in range(4)]
    try:
        for c, lo, hi in zip(ctxs, cuts[:-1], cuts[1:]):
            c.set_barcodes(table, 1)
            if hi > lo:
                c.demux_assign(np.ascontiguousarray(bc[lo:hi]))
        partial = [c.counts() for c in ctxs]
        ctxs[0].counts_allreduce(ctxs[1:])
        expect = oracle.demux_batch(table, bc, 1)[4]
        assert np.array_equal(sum(partial), expect)
        for c in ctxs:                                    # every ctx ends with the totals
            assert np.array_equal(c.counts(), expect)
        assert int(expect[:96].sum()) == int(expect[97]) and int(expect[96]) == n
    finally:
        for c in ctxs:
            c.close()


def test_counts_allreduce_rejects_mismatched_sheets(hip_lib):
    import seqkit_amd
    a, b = seqkit_amd.Context(0), seqkit_amd.Context(0)
    try:
        a.set_barcodes(synth.make_sheet(16, 8, seed=3), 1)
        b.set_barcodes(synth.make_sheet(8, 8, seed=3), 1)
        with pytest.raises(seqkit_amd.SeqkitHipError):
            a.counts_allreduce([b])
    finally:
        a.close(); b.close()


def test_rank_communicator_of_one(hip_lib, oracle):
    """ncclGetUniqueId + ncclCommInitRank(1 rank) + ncclAllReduce on the ctx stream: the sum over one rank is the identity."""
    import seqkit_amd
    table = synth.make_sheet(16, 8, seed=3)
    bc, _ = synth.observe_barcodes(table, 5000, seed=5)
    with seqkit_amd.Context(0) as c:
        c.set_barcodes(table, 1)
        uid = capi.comm_unique_id()
        assert len(uid) == 128
        c.comm_init_rank(uid, 0, 1)
        c.demux_assign(bc)
        before = c.counts()
        c.counts_allreduce()
        c.sync()
        assert np.array_equal(c.counts(), before) and np.array_equal(before, oracle.demux_batch(table, bc, 1)[4])
        # a caller-owned device vector (BAM counters + histogram shape)
        h = np.arange(5004, dtype=np.uint64)
        d = c.malloc_device(h.nbytes)
        c.copy_h2d(d, h)
        c.allreduce_u64_dev(d, h.size)
        back = np.zeros_like(h)
        c.copy_d2h(back, d)
        c.sync()
        c.free_device(d)
        assert np.array_equal(back, h)
        with pytest.raises(seqkit_amd.SeqkitHipError):
            c.comm_init_rank(uid, 0, 1)                   # a ctx has one communicator
        c.comm_destroy()
