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


def test_placement_tuning_keeps_the_bytes_and_the_counters(hip_lib, oracle):
    """sk_fused_tune_placement_dev: K candidate device buffers per matrix; whatever combination it settles on, the pass over
    it gives the oracle's bytes, the chosen pointers are candidates, and the ctx counters were not touched by the probes."""
    import seqkit_amd
    n, L, K = 20000, 150, 3
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=91, halves=2)
    mates_h = []
    for mi in range(2):
        seq, qual = synth.make_reads(n, L, seed=310 + mi)
        mates_h.append((seq, synth.add_forced_classes(qual, seed=320 + mi)))
    with seqkit_amd.Context(0) as c:
        c.set_barcodes(table, 1)
        dev = []

        def dmalloc(nbytes, src=None):
            p = c.malloc_device(nbytes + 16)
            dev.append(p)
            if src is not None:
                c.copy_h2d(p, np.ascontiguousarray(src))
            return p
        cands = [{"seq": [dmalloc(n * L, mates_h[i][0]) for _ in range(K)], "qual": [dmalloc(n * L, mates_h[i][1]) for _ in range(K)],
                  "out_seq": [dmalloc(n * L) for _ in range(K)]} for i in range(2)]
        d_bc, d_assign = dmalloc(n * 17, bc), dmalloc(n * 4)
        d_lk = [dmalloc(n * 2) for _ in range(2)]
        mates = [{"seq": cands[i]["seq"][0], "qual": cands[i]["qual"][0], "len": 0, "out_seq": cands[i]["out_seq"][0], "lowest_k": d_lk[i]} for i in range(2)]
        c.sync()
        before = c.counts()
        chosen, ms0, ms1, probes = c.fused_tune_placement_dev(n, L, 20, mates, cands, bc=d_bc, bc_stride=17, assign=d_assign, sweeps=2)
        assert np.array_equal(c.counts(), before) and int(before.sum()) == 0
        assert probes >= 1 + 6 * (K - 1) and ms0 > 0 and 0 < ms1 <= ms0 * 1.0001
        for i in range(2):
            assert chosen[i]["seq"] in cands[i]["seq"] and chosen[i]["qual"] in cands[i]["qual"] and chosen[i]["out_seq"] in cands[i]["out_seq"]
            assert chosen[i]["lowest_k"] == d_lk[i]
        c.fused_pass_dev(n, L, 20, chosen, bc=d_bc, bc_stride=17, assign=d_assign)
        got_assign = np.empty(n, dtype=np.int32)
        c.copy_d2h(got_assign, d_assign)
        outs = [np.empty((n, L), dtype=np.uint8) for _ in range(2)]
        lks = [np.empty(n, dtype=np.uint16) for _ in range(2)]
        for i in range(2):
            c.copy_d2h(outs[i], chosen[i]["out_seq"])
            c.copy_d2h(lks[i], d_lk[i])
        c.sync()
        e = oracle.demux_batch(table, bc, 1)
        assert np.array_equal(got_assign, e[0]) and np.array_equal(c.counts(), e[4])
        for i in range(2):
            assert np.array_equal(outs[i], oracle.mask_batch(mates_h[i][0], mates_h[i][1], None, 20))
            assert np.array_equal(lks[i], oracle.trim_batch(mates_h[i][1], None, 20))
        # bad arguments are codes
        with pytest.raises(seqkit_amd.SeqkitHipError):
            c.fused_tune_placement_dev(n, L, 20, mates, [dict(cands[0], qual=[cands[0]["qual"][0], 0, 0]), cands[1]], bc=d_bc, bc_stride=17, assign=d_assign)
        for p in dev:
            c.free_device(p)


def test_bench_two_ranks_on_one_gpu_take_the_gloo_fallback(hip_lib, oracle):
    """The whole N > 1 flow of bench.py on a one-GPU box: its launcher starts two ranks, both are put on device 0
    (SK_BENCH_SAME_DEVICE), RCCL refuses two ranks on one device, ALL ranks fall back to the gloo sum together, the counter
    identities over both shards hold (asserted inside bench.py) and rank 0's line says n_gpus = 2 and which reduce ran."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SK_BENCH_SAME_DEVICE="1", SK_BENCH_LAUNCH_TIMEOUT="500")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--pairs", "2000000", "--placements", "2", "--steps", "3", "--warmup", "1",
                        "--no-extra", "--faithful-reads", "0", "--cpu-sample", "200000"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity_sample_ok"] is True and d["scaling"] == "weak"
    assert "gloo" in d["config"]["count_reduce"] or "RCCL" in d["config"]["count_reduce"]
    assert d["value"] > 0 and d["config"]["placement"]["candidates"] == 2


def _device_count():
    lib = capi.load_library()
    return int(lib.sk_device_count())


needs_two = pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X in one process (the builder's GPU boxes have one; a node with more runs it)")


@needs_two
def test_demultiplex_by_table_on_a_second_device_while_the_first_is_current(hip_lib, oracle):
    """The sheet's lookup table is uploaded on first use: that upload must land on the ctx's device even though the calling
    thread's current device is another (round 2's advisor: ensure_neighbour_table ran before the ctx was bound)."""
    import seqkit_amd
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, 50_000, seed=9, halves=2)
    with seqkit_amd.Context(0) as c0, seqkit_amd.Context(1) as c1:
        c0.set_barcodes(table, 1)
        c0.demux_assign(bc, want_detail=False)              # leaves device 0 current on this thread
        c1.set_barcodes(table, 1)
        assign, *_ = c1.demux_assign(bc, want_detail=False)
        e = oracle.demux_batch(table, bc, 1)
        assert np.array_equal(assign, e[0]) and np.array_equal(c1.counts(), e[4])
        c1.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
        assign, low, first, last = c1.demux_assign(bc)
        m = e[0] != -1
        assert np.array_equal(assign, e[0]) and np.array_equal(low[m], e[1][m]) and np.array_equal(first[m], e[2][m]) and np.array_equal(last[m], e[3][m])


@needs_two
def test_counts_allreduce_between_two_devices(hip_lib, oracle):
    """sk_counts_allreduce over ctxs on DISTINCT devices: ncclCommInitAll over the device list and one grouped ncclAllReduce
    on the ctx streams (the leg a one-GPU box cannot run), with a second ctx on each device summed on that device first."""
    import seqkit_amd
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    n = 40_000
    bc, _ = synth.observe_barcodes(table, n, seed=78, halves=2)
    ctxs = [seqkit_amd.Context(d) for d in (0, 1, 0, 1)]
    cuts = [0, 9000, 21000, 21000, n]
    try:
        for c, lo, hi in zip(ctxs, cuts[:-1], cuts[1:]):
            c.set_barcodes(table, 1)
            if hi > lo:
                c.demux_assign(np.ascontiguousarray(bc[lo:hi]), want_detail=False)
        ctxs[0].counts_allreduce(ctxs[1:])
        expect = oracle.demux_batch(table, bc, 1)[4]
        for c in ctxs:
            assert np.array_equal(c.counts(), expect)
        ctxs[0].counts_allreduce(ctxs[1:])                  # again on the cached communicators: every ctx now holds 4 x the totals
        assert np.array_equal(ctxs[3].counts(), 4 * expect)
    finally:
        for c in ctxs:
            c.close()
