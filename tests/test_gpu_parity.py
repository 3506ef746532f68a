"""GPU: the HIP path, called through the C-ABI, against the CPU oracle — bit-exact (all integer/byte work)."""
import os

import numpy as np
import pytest

from seqkit_amd import synth

pytestmark = pytest.mark.gpu


def b(s: str) -> bytes:
    return s.encode("latin-1")


def rows(strings, stride=None):
    L = max(len(s) for s in strings)
    stride = stride or max(L, 1)
    m = np.full((len(strings), stride), ord("~"), dtype=np.uint8)
    ln = np.zeros(len(strings), dtype=np.uint16)
    for i, s in enumerate(strings):
        m[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
        ln[i] = len(s)
    return m, ln


# ---- golden vectors through the GPU ------------------------------------------------------------------
def test_trim_kat_gpu(ctx, golden):
    g = golden["trim_by_quality"]
    qual, ln = rows([b(c["qual"]) for c in g["cases"]])
    got = ctx.trim_by_quality(qual, ln, g["min_baseq"])
    assert list(map(int, got)) == [c["lowest_k"] for c in g["cases"]]


def test_mask_kat_gpu(ctx, golden):
    for c in golden["mask_by_quality"]["cases"]:
        seq, ln = rows([b(c["seq"])])
        qual, _ = rows([b(c["qual"])])
        out = ctx.mask_by_quality(seq, qual, ln, c["min_baseq"])
        assert out[0, :ln[0]].tobytes() == b(c["out"]), c


def test_demux_kat_gpu(ctx, golden):
    g = golden["demultiplex"]
    for c in g["cases"]:
        table = np.array([list(b(x)) for x in c["sheet"]], dtype=np.uint8)
        ctx.set_barcodes(table, g["max_diff"])
        bc = np.array([list(b(c["observed"]))], dtype=np.uint8)
        assign, low, first, last = ctx.demux_assign(bc)
        assert (int(low[0]), int(first[0]), int(last[0]), int(assign[0])) == (c["lowest_diff"], c["first"], c["last"], c["assign"]), c


def test_bam_kat_gpu(ctx, golden):
    g = golden["bam"]
    counters, hist, total = ctx.bam_flag_tlen(np.array(g["flags"], dtype=np.uint16), np.array(g["tid"], dtype=np.int32),
                                              np.array(g["mtid"], dtype=np.int32), np.array(g["tlen"], dtype=np.int32), 5000)
    assert list(map(int, counters)) == [g["total"], g["aligned"], g["duplicate"]]
    assert {str(i): int(hist[i]) for i in np.nonzero(hist)[0]} == g["hist_nonzero"] and total == 1


# ---- trim -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("min_baseq", [0, 2, 20, 30, 41, 255])
def test_trim_cfg2_classes(ctx, oracle, min_baseq):
    """cfg 2 shape (150 bp, forced classes) at every threshold of SURVEY.md §8(d)."""
    _, qual = synth.make_reads(20000, 150, seed=2)
    qual = synth.add_forced_classes(qual, seed=2)
    got = ctx.trim_by_quality(qual, None, min_baseq)
    assert np.array_equal(got, oracle.trim_batch(qual, None, min_baseq))


@pytest.mark.parametrize("stride,n", [(1, 70), (3, 129), (37, 1000), (150, 4097), (151, 333), (250, 2000), (960, 130), (1000, 70), (2100, 65)])
def test_trim_ragged_random_bytes(ctx, oracle, stride, n):
    """Any stride (LDS tile path up to 960, row-per-thread path above), ragged lengths incl. 0 and 1, arbitrary bytes."""
    rng = np.random.default_rng(stride * 1000 + n)
    qual = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    ln = synth.ragged_lengths(n, stride, seed=stride)
    for m in (0, 7, 20, 200, 255):
        got = ctx.trim_by_quality(qual, ln, m)
        assert np.array_equal(got, oracle.trim_batch(qual, ln, m)), (stride, n, m)
    got = ctx.trim_by_quality(qual, None, 20)
    assert np.array_equal(got, oracle.trim_batch(qual, None, 20))


def test_trim_low_entropy_ties(ctx, oracle):
    """Many exact ties and zero increments: the rightmost-minimum rule and the early break."""
    rng = np.random.default_rng(5)
    qual = (rng.integers(0, 3, size=(5000, 150)) * 1 + 33 + 19).astype(np.uint8)     # Q19..Q21 around the threshold
    assert np.array_equal(ctx.trim_by_quality(qual, None, 20), oracle.trim_batch(qual, None, 20))
    qual[:] = 33 + 20
    assert np.array_equal(ctx.trim_by_quality(qual, None, 20), oracle.trim_batch(qual, None, 20))


@pytest.mark.parametrize("stride", [8, 9, 15, 16, 17, 24, 31, 40, 150, 250])
def test_trim_long_scans_every_step_count(ctx, oracle, stride):
    """Scans that never break, over rows of an even and an odd number of whole 8-byte steps (the scan loop takes two steps
    per iteration), alone in a tile and among rows that break early (hand-over of the last rows to the whole wave)."""
    rng = np.random.default_rng(stride)
    n = 64 * 5 + 7
    qual = np.full((n, stride), ord("#"), dtype=np.uint8)
    early = rng.random(n) < 0.9
    early[:64] = False                                   # a whole tile of rows that go to the end
    early[64:128] = True                                 # one of rows that all break early
    qual[early] = (rng.integers(25, 41, size=(int(early.sum()), stride)) + 33).astype(np.uint8)
    for ln in (None, synth.ragged_lengths(n, stride, seed=stride)):
        for m in (20, 2):
            assert np.array_equal(ctx.trim_by_quality(qual, ln, m), oracle.trim_batch(qual, ln, m)), (stride, m, ln is None)


@pytest.mark.parametrize("stride", [150, 37, 251])
def test_trim_alone_bytes_below_33(ctx, oracle, stride):
    """Trim alone scans the raw bytes (m + 33 for m) and checks on the way that no byte of the row was below '!': one such
    byte at every position of a row — in a whole step, in the step the row stops in, in the row's partial last step, in
    the part the whole wave finishes — in rows that break early and rows that never do; such rows take the byte-wise path."""
    rng = np.random.default_rng(stride + 1)
    rows_ = []
    for base in ("#", "read"):
        for pos in range(stride):
            q = (np.full(stride, ord("#"), dtype=np.uint8) if base == "#"
                 else (np.clip(rng.normal(30, 6, size=stride).round(), 2, 40) + 33).astype(np.uint8))
            q[pos] = rng.integers(0, 33)
            rows_.append(q)
    qual = np.stack(rows_)
    rng.shuffle(qual, axis=0)
    clean = np.full((64 * 3, stride), ord("#"), dtype=np.uint8)          # tiles without such a byte around them
    qual = np.concatenate([clean[:64], qual, clean[64:]])
    for ln in (None, synth.ragged_lengths(len(qual), stride, seed=stride)):
        for m in (20, 0, 255):
            assert np.array_equal(ctx.trim_by_quality(qual, ln, m), oracle.trim_batch(qual, ln, m)), (stride, m, ln is None)


def test_trim_empty_batch(ctx):
    assert ctx.trim_by_quality(np.zeros((0, 150), dtype=np.uint8), None, 20).shape == (0,)


# ---- mask -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("min_baseq", [0, 1, 20, 95, 96, 222, 223, 224, 255])
def test_mask_all_threshold_modes(ctx, oracle, min_baseq):
    """Every packed-compare mode (none / [33,33+m) / q>=33 / wrapped interval), every byte value."""
    rng = np.random.default_rng(min_baseq)
    n, stride = 3000, 150
    seq = synth.BASES[rng.integers(0, 4, size=(n, stride))]
    qual = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    qual[0, :] = np.arange(150, dtype=np.uint8)
    qual[1, :106] = np.arange(150, 256, dtype=np.uint8)
    got = ctx.mask_by_quality(seq, qual, None, min_baseq)
    assert np.array_equal(got, oracle.mask_batch(seq, qual, None, min_baseq))


@pytest.mark.parametrize("n,stride", [(1, 1), (1, 15), (3, 17), (64, 150), (65, 150), (1000, 33), (10000, 150), (70, 1000)])
def test_mask_cfg1_shapes(ctx, oracle, n, stride):
    """cfg 1 (10 k x 150 bp, min_baseq 20) and odd sizes whose byte count is not a multiple of 16."""
    seq, qual = synth.make_reads(n, stride, seed=1)
    got = ctx.mask_by_quality(seq, qual, None, 20)
    assert np.array_equal(got, oracle.mask_batch(seq, qual, None, 20))


# ---- demultiplex -------------------------------------------------------------------------------------------
def check_demux(ctx, oracle, table, bc, max_diff=1):
    ctx.set_barcodes(table, max_diff)
    assign, low, first, last = ctx.demux_assign(bc)
    e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, max_diff)
    assert np.array_equal(assign, e_assign)
    assert np.array_equal(low, e_low)
    assert np.array_equal(first, e_first)
    assert np.array_equal(last, e_last)
    assert np.array_equal(ctx.counts(), e_counts)
    return e_counts


@pytest.fixture(params=["bitsliced", "onehot", "bytes"])
def demux_path(request, monkeypatch):
    """Force each of the three matcher implementations (bit-sliced planes / one-hot popcount / byte compare)."""
    if request.param in ("onehot", "bytes"):
        monkeypatch.setenv("SK_NO_BITSLICE", "1")
    if request.param == "bytes":
        monkeypatch.setenv("SK_NO_ONEHOT", "1")
    return request.param


def test_demux_all_matchers_agree(ctx, oracle, demux_path):
    for S, dual, seed in ((16, False, 3), (96, True, 4), (33, False, 5), (150, True, 6)):
        table = synth.make_sheet(S, 8, dual=dual, seed=seed)
        bc, _ = synth.observe_barcodes(table, 20011, seed=seed, halves=2 if dual else 1)
        bc[::7, 3] = ord("N")
        bc[::11, 0] = ord("x")
        check_demux(ctx, oracle, table, bc)
    # wildcards, duplicates, '+' and a lone sample
    table = np.array([list(b"ACGTNNGT+AAUU"), list(b"ACGTACGT+AAUU"), list(b"TTTTTTTT+CCCC"), list(b"TTTTTTTT+CCCC")], dtype=np.uint8)
    rng = np.random.default_rng(77)
    bc = table[rng.integers(0, 4, size=9000)].copy()
    hit = rng.random(9000) < 0.7
    bc[hit, rng.integers(0, 13, size=9000)[hit]] = np.frombuffer(b"ACGTN+U", dtype=np.uint8)[rng.integers(0, 7, size=int(hit.sum()))]
    check_demux(ctx, oracle, table, bc)
    check_demux(ctx, oracle, table[:1], bc, max_diff=3)


def test_demux_many_samples_generic_groups(ctx, oracle):
    """S = 700 (22 groups of 32: the generic group loop of the bit-sliced matcher) and S = 2000 (tables too big for LDS)."""
    rng = np.random.default_rng(88)
    for S in (700, 2000):
        table = synth.BASES[rng.integers(0, 4, size=(S, 10))]
        bc = table[rng.integers(0, S, size=30000)].copy()
        hit = rng.random(30000) < 0.5
        bc[hit, rng.integers(0, 10, size=30000)[hit]] = synth.BASES[rng.integers(0, 4, size=int(hit.sum()))]
        check_demux(ctx, oracle, table, bc)


def test_demux_cfg3_single_index(ctx, oracle):
    table = synth.make_sheet(16, 8, dual=False, seed=3)
    bc, _ = synth.observe_barcodes(table, 200000, seed=3)
    counts = check_demux(ctx, oracle, table, bc)
    S = 16
    assert counts[:S].sum() == counts[S + 1] and counts[S] == 200000


def test_demux_cfg4_dual_index(ctx, oracle):
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    assert table.shape == (96, 17)
    bc, _ = synth.observe_barcodes(table, 100001, seed=4, halves=2)
    counts = check_demux(ctx, oracle, table, bc)
    assert counts[:96].sum() == counts[97]


def test_demux_ambiguity_wildcards_and_umi_columns(ctx, oracle):
    """cfg 3 parity sheet: two barcodes at distance 2 (forces ambiguity), an N column, a UUUU block, duplicates."""
    table = np.array([list(b"ACGTACGTAAAA"), list(b"ACGTACGAAAAT"), list(b"TTNTTTTTUUUU"), list(b"GGGGGGGGUUUU"),
                      list(b"CCCCCCCCCCCC"), list(b"CCCCCCCCCCCC")], dtype=np.uint8)
    rng = np.random.default_rng(9)
    src = table[rng.integers(0, 6, size=50000)].copy()
    alphabet = np.frombuffer(b"ACGTNU+acgt", dtype=np.uint8)
    for _ in range(2):
        hit = rng.random(src.shape[0]) < 0.5
        pos = rng.integers(0, 12, size=src.shape[0])
        src[hit, pos[hit]] = alphabet[rng.integers(0, alphabet.size, size=int(hit.sum()))]
    counts = check_demux(ctx, oracle, table, src)
    assert counts[6 + 2] > 0          # ambiguity did occur


@pytest.mark.parametrize("max_diff", [0, 2, 17])
def test_demux_other_max_diff(ctx, oracle, max_diff):
    table = synth.make_sheet(16, 8, dual=False, seed=11)
    bc = synth.BASES[np.random.default_rng(11).integers(0, 4, size=(30000, 8))]
    check_demux(ctx, oracle, table, bc, max_diff)


def test_demux_byte_path_many_classes(ctx, oracle):
    """More than 7 distinct sheet bytes at a position -> byte-for-byte matcher; also arbitrary bytes and a long barcode."""
    rng = np.random.default_rng(12)
    table = rng.integers(32, 127, size=(40, 9), dtype=np.uint8)
    bc = table[rng.integers(0, 40, size=20000)].copy()
    hit = rng.random(20000) < 0.6
    bc[hit, rng.integers(0, 9, size=20000)[hit]] = rng.integers(0, 256, size=int(hit.sum()), dtype=np.uint8)
    check_demux(ctx, oracle, table, bc)
    table = synth.BASES[rng.integers(0, 4, size=(10, 40))]            # L = 40 > one-hot limit
    bc = table[rng.integers(0, 10, size=5000)].copy()
    bc[:, 5] = ord("A")
    check_demux(ctx, oracle, table, bc)


def test_demux_strided_rows_and_tail(ctx, oracle):
    """bc_stride > L (padded rows) and a row count that is not a multiple of the 64-row tile."""
    table = synth.make_sheet(16, 8, dual=False, seed=13)
    bc8, _ = synth.observe_barcodes(table, 777, seed=13)
    bc = np.full((777, 12), ord("#"), dtype=np.uint8)
    bc[:, :8] = bc8
    ctx.set_barcodes(table, 1)
    assign, low, first, last = ctx.demux_assign(bc)
    e = oracle.demux_batch(table, bc8, 1)
    assert np.array_equal(assign, e[0]) and np.array_equal(low, e[1]) and np.array_equal(first, e[2]) and np.array_equal(last, e[3])


def test_demux_single_sample_and_empty_sheet(ctx, oracle):
    table = np.array([list(b"ACGTACGT")], dtype=np.uint8)
    bc = synth.BASES[np.random.default_rng(14).integers(0, 4, size=(1000, 8))]
    bc[:10] = table[0]
    check_demux(ctx, oracle, table, bc)
    empty = np.zeros((0, 8), dtype=np.uint8)
    ctx.set_barcodes(empty, 1)
    assign, low, first, last = ctx.demux_assign(bc)
    e = oracle.demux_batch(empty, bc, 1)
    assert np.array_equal(assign, e[0]) and (assign == -1).all() and np.array_equal(low, e[1])


def test_counters_accumulate_and_reset(ctx, oracle):
    table = synth.make_sheet(16, 8, dual=False, seed=15)
    bc, _ = synth.observe_barcodes(table, 5000, seed=15)
    ctx.set_barcodes(table, 1)
    ctx.demux_assign(bc)
    ctx.demux_assign(bc)
    e = oracle.demux_batch(table, bc, 1)[4]
    assert np.array_equal(ctx.counts(), 2 * e)
    ctx.counts_reset()
    assert ctx.counts().sum() == 0


# ---- demultiplex alone through the neighbourhood table (demux_lut_kernel): decision only, or detail of matched rows ----
def check_demux_decision_only(ctx, oracle, table, bc, max_diff=1):
    """assign + counters without the detail columns: served by the lookup table when the sheet has one, by the matchers
    otherwise — the same answers either way, and the same as with the detail columns."""
    ctx.set_barcodes(table, max_diff)
    assign, low, first, last = ctx.demux_assign(bc, want_detail=False)
    e_assign, _, _, _, e_counts = oracle.demux_batch(table, bc, max_diff)
    assert low is None and np.array_equal(assign, e_assign)
    assert np.array_equal(ctx.counts(), e_counts)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["16 x 8", "96 x 8+8", "200 x 8+8"])
@pytest.mark.parametrize("n", [3000, 700_000])
@pytest.mark.parametrize("table_in", ["LDS", "vector cache"])
def test_demux_by_table_into_the_callers_counters(ctx, oracle, monkeypatch, shape, n, table_in):
    """The lookup kernels leave `identified` to whoever sums the histogram: the fold kernel behind a launch of thousands of
    workgroups, the workgroup itself behind a small one that adds to the caller's vector directly (sk_kernels.hip,
    lut_identified_from_hist).  Both, twice into the same vector, against the oracle's counters.  (A table in LDS means one
    workgroup of 16 waves per CU: only the form that reads its table through the vector cache launches thousands.)"""
    from seqkit_amd import synth
    if table_in == "vector cache":
        monkeypatch.setenv("SK_DEMUX_LDSTAB", "0")
    S = int(shape.split(" x ")[0])
    dual = "+" in shape
    table = synth.make_sheet(S, 8, dual=dual, seed=S)
    bc, _ = synth.observe_barcodes(table, n, seed=S + 1, halves=2 if dual else 1)
    ctx.set_barcodes(table, 1)
    ctx.counts_reset()
    zeros = np.zeros(S + 3, dtype=np.uint64)
    d_bc, d_assign, d_counts = ctx.malloc_device(bc.nbytes), ctx.malloc_device(4 * n), ctx.malloc_device(zeros.nbytes)
    try:
        ctx.copy_h2d(d_bc, bc)
        ctx.copy_h2d(d_counts, zeros)
        for _ in range(2):
            ctx.demux_assign_dev(d_bc, bc.shape[1], n, d_assign, counts=d_counts)
        ctx.sync()
        assign, got = np.empty(n, dtype=np.int32), np.empty(S + 3, dtype=np.uint64)
        ctx.copy_d2h(assign, d_assign)
        ctx.copy_d2h(got, d_counts)
    finally:
        for p in (d_bc, d_assign, d_counts):
            ctx.free_device(p)
    e_assign, _, _, _, e_counts = oracle.demux_batch(table, bc, 1)
    assert np.array_equal(assign, e_assign)
    assert np.array_equal(got, 2 * e_counts.astype(np.uint64)), (got[S:], 2 * e_counts[S:])
    assert ctx.counts().sum() == 0                      # nothing went to the ctx's own counters


def check_demux_matched(ctx, oracle, table, bc, max_diff=1):
    """SK_DETAIL_MATCHED: assign and counters of every row, lowest_diff / first / last of the rows that matched something
    (the only rows the reference reads them for: src/fasta_demultiplex.rs:184-188)."""
    import seqkit_amd
    ctx.set_barcodes(table, max_diff)
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
    try:
        assign, low, first, last = ctx.demux_assign(bc)
    finally:
        ctx.set_detail_mode(seqkit_amd.SK_DETAIL_FULL)
    e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, max_diff)
    assert np.array_equal(assign, e_assign)
    m = e_assign != -1
    assert np.array_equal(low[m], e_low[m]) and np.array_equal(first[m], e_first[m]) and np.array_equal(last[m], e_last[m])
    assert np.array_equal(ctx.counts(), e_counts)
    return int(m.sum())


@pytest.fixture(params=["default", "no table", "aligned rows gathered too", "table in the vector cache", "two rows per lane", "one row per lane",
                        "never half by half"])
def lut_form(request, monkeypatch):
    """The forms of the lookup kernel (and the matchers, without a table) on the same inputs."""
    if request.param == "no table":
        monkeypatch.setenv("SK_NO_HASH_DEMUX", "1")
    elif request.param == "never half by half":
        monkeypatch.setenv("SK_DEMUX_PAIR", "0")            # a table too large for the LDS is then probed through the vector cache
    elif request.param == "two rows per lane":
        monkeypatch.setenv("SK_DEMUX_ROWS2", "1")
    elif request.param == "one row per lane":
        monkeypatch.setenv("SK_DEMUX_ROWS2", "0")
    elif request.param == "aligned rows gathered too":
        monkeypatch.setenv("SK_DEMUX_DIRECT", "0")
    elif request.param == "table in the vector cache":
        monkeypatch.setenv("SK_DEMUX_LDSTAB", "0")
    return request.param


def test_demux_by_table_cfg3_cfg4(ctx, oracle, lut_form):
    table = synth.make_sheet(16, 8, dual=False, seed=3)
    bc, _ = synth.observe_barcodes(table, 200_003, seed=3)
    bc[::13, 5] = ord("N")
    bc[::17, 2] = ord("+")
    check_demux_decision_only(ctx, oracle, table, bc)
    check_demux_decision_only(ctx, oracle, table, bc, max_diff=0)
    assert check_demux_matched(ctx, oracle, table, bc) > 100_000
    check_demux_matched(ctx, oracle, table, bc, max_diff=0)
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, 100_001, seed=4, halves=2)
    bc[::19, 8] = ord("A")                                          # a broken separator costs one mismatch for every sample
    bc[::23, 8] = ord("-")
    bc[::29, 3] = ord("+")
    check_demux_decision_only(ctx, oracle, table, bc)
    check_demux_decision_only(ctx, oracle, table, bc, max_diff=0)
    assert check_demux_matched(ctx, oracle, table, bc) > 50_000
    check_demux_matched(ctx, oracle, table, bc, max_diff=0)
    for stride in (20, 24, 32):                                     # bc_stride above the barcode length
        padded = np.full((bc.shape[0], stride), 0x41, dtype=np.uint8)
        padded[:, :17] = bc
        ctx.set_barcodes(table, 1)
        assign, *_ = ctx.demux_assign(padded, want_detail=False)
        assert np.array_equal(assign, oracle.demux_batch(table, bc, 1)[0])


@pytest.mark.parametrize("S,dual", [(129, False), (384, True), (1000, True), (1000, False), (1021, False)])
def test_demux_by_table_many_samples(ctx, oracle, lut_form, S, dual):
    """Sheets of more than 128 samples take the lookup too (sk_lut.h): sample indices of 10 bits; a dual-index sheet whose
    full-key table would not fit the LDS is looked up half by half (384 = 24 x 16 combinations: two small tables and the
    pair table), with duplicates (always ambiguous), a broken separator, foreign bytes and a UMI column behind it."""
    rng = np.random.default_rng(S)
    if S >= 1000 and not dual:                                       # (a thousand 8-mers three apart take the greedy generator minutes: random 12- / 10-mers)
        table = np.unique(synth.BASES[rng.integers(0, 4, size=(2 * S, 12 if S == 1000 else 10))], axis=0)
        table = np.ascontiguousarray(table[rng.permutation(table.shape[0])[:S]])
    else:
        table = synth.make_sheet(S, 8, dual=dual, seed=S)
    if S >= 384:
        table[7] = table[3]                                          # duplicate rows: ambiguous whatever the read
        table[S - 1] = table[S // 2]
    n = 120_001 if S < 384 else (40_001 if S == 384 else 12_001)   # (the oracle's loop is S x n x L on the host)
    bc, _ = synth.observe_barcodes(table, n, seed=S + 1, halves=2 if dual else 1)
    bc[::13, 5] = ord("N")
    bc[::29, 3] = ord("+")
    if dual:
        bc[::19, 8] = ord("A")                                       # a broken separator costs one mismatch for every sample
    check_demux_decision_only(ctx, oracle, table, bc)
    check_demux_decision_only(ctx, oracle, table, bc, max_diff=0)
    assert check_demux_matched(ctx, oracle, table, bc) > n // 3
    check_demux_matched(ctx, oracle, table, bc, max_diff=0)
    e = oracle.demux_batch(table, bc, 1)
    if S >= 384:
        assert (e[0] == -2).sum() > 20
    if dual:                                                         # ... and with a UMI column behind the second index
        t2 = np.concatenate([table, np.full((S, 3), ord("U"), dtype=np.uint8)], axis=1)
        b2 = np.concatenate([bc, rng.choice(synth.BASES, size=(bc.shape[0], 3))], axis=1)
        check_demux_matched(ctx, oracle, np.ascontiguousarray(t2), np.ascontiguousarray(b2))


@pytest.mark.parametrize("shape", ["16 x 8", "96 x 8+8", "384 x 8+8", "40 x 4+4", "24 x 6"])
def test_demux_by_table_mixed_case_sheets(ctx, oracle, lut_form, shape):
    """The reference compares raw bytes (src/fasta_demultiplex.rs:273-274), so a sheet may be typed partly in lower case: eight
    letters and more, where the table's 3-bit classes hold seven.  Such sheets get 4-bit classes (sk_lut.h, "Wide classes") when
    a row — or each half beside the separator — is at most 8 columns: the full-key table for the single-index shapes, the
    factored form for the dual-index ones.  Reads in both cases, `N`s, a broken separator, foreign bytes; every form of the
    lookup kernels; the table really is what runs (sk_barcode_table_info)."""
    from seqkit_amd import capi
    S = int(shape.split(" x ")[0])
    cols = shape.split(" x ")[1]
    dual = "+" in cols
    hl = int(cols.split("+")[0])
    rng = np.random.default_rng(S + hl)
    if hl == 8:
        table = synth.make_sheet(S, 8, dual=dual, seed=S + 5)
    else:                                                          # short barcodes: distinct random rows
        half = lambda k: np.unique(synth.BASES[rng.integers(0, 4, size=(4 * k, hl))], axis=0)[:k]

        def apart(k):                                              # k rows pairwise at least 3 apart (what the factored form needs of its halves)
            out = []
            while len(out) < k:
                c = synth.BASES[rng.integers(0, 4, size=hl)]
                if all((c != o).sum() >= 3 for o in out):
                    out.append(c)
            return np.array(out, dtype=np.uint8)
        if dual:
            a, b = apart(7), apart(7)
            combos = rng.permutation(len(a) * len(b))[:S]
            table = np.concatenate([a[combos % len(a)], np.full((S, 1), ord("+"), dtype=np.uint8), b[combos // len(a)]], axis=1)
        else:
            table = half(S)
        table = np.ascontiguousarray(table.astype(np.uint8))
    lower = lambda x: np.where((x >= 65) & (x <= 90), x + 32, x).astype(np.uint8)
    table[1::2] = lower(table[1::2])                               # every other sample in lower case
    if not dual:
        table[0, 0] = ord("N")                                     # a wildcard of ONE row (entered once per class of that column): only without a separator
    n = 60_001
    src = table.copy()
    src[src == ord("N")] = ord("A")
    bc, _ = synth.observe_barcodes(src, n, seed=S + 9, halves=2 if dual else 1)
    flip = rng.random(n) < 0.3                                     # a third of the reads in the other case
    bc[flip] = np.where((bc[flip] >= 97) & (bc[flip] <= 122), bc[flip] - 32, lower(bc[flip])).astype(np.uint8)
    bc[::13, 1] = ord("N")
    bc[::31, 2] = ord("#")
    if dual:
        bc[::19, hl] = ord("a")                                    # a broken separator costs one mismatch for every sample
    # ONE byte repeated over a whole half (or row) beside a valid other half: eight bytes of class 15 pack to 0xFFFFFFFF, the word a
    # free slot of the factored form held alone until round 6 — a poly-G index read got the sample of half 0
    probe_bytes = b"ACGTNacgtn+#\x00\xff"
    for j, b in enumerate(probe_bytes):
        for part in range(3 if dual else 1):
            r = 1000 + 3 * j + part
            bc[r] = table[(7 * j + part) % S]
            bc[r][bc[r] == ord("N")] = ord("A")
            lo, hi = (0, hl) if part == 0 else ((hl + 1, 2 * hl + 1) if part == 1 else (0, bc.shape[1]))
            keep_sep = bc[r, hl] if dual else None
            bc[r, lo:hi] = b
            if dual:
                bc[r, hl] = keep_sep
    ctx.set_barcodes(table, 1)
    kind = ctx.barcode_table_info()["kind"]
    if lut_form == "no table" or (dual and lut_form == "never half by half"):      # (wide classes beside a separator: the factored form or the matchers)
        assert kind == capi.SK_TABLE_NONE, kind
    else:
        assert kind & capi.SK_TABLE_WIDE_CLASSES, kind
        assert (kind & 3) == (capi.SK_TABLE_FACTORED if dual else capi.SK_TABLE_FULL_KEY), kind
    check_demux_decision_only(ctx, oracle, table, bc)
    check_demux_decision_only(ctx, oracle, table, bc, max_diff=0)
    assert check_demux_matched(ctx, oracle, table, bc) > n // 4
    check_demux_matched(ctx, oracle, table, bc, max_diff=0)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 511, 512, 513, 1023, 1025])
def test_demux_by_table_around_the_quad(ctx, oracle, lut_form, n):
    """The lookup kernels' unit is 256 rows whose codes (and detail columns) change lanes through LDS and leave as wide stores; a
    call's last, partial unit writes its narrow columns row by row (sk_kernels.hip): row counts around every boundary of the
    64-row tile, the 128-row tile of the two-rows-per-lane kernel and the 256-row unit, on both benchmark sheets, decision alone
    and with the detail columns of matched rows; nothing is written behind row n."""
    for S, dual in ((16, False), (96, True)):
        table = synth.make_sheet(S, 8, dual=dual, seed=S)
        bc, _ = synth.observe_barcodes(table, n, seed=n, halves=2 if dual else 1)
        check_demux_decision_only(ctx, oracle, table, bc)
        check_demux_matched(ctx, oracle, table, bc)
        # device entry point with guard rows behind the outputs
        ctx.set_barcodes(table, 1)
        from seqkit_amd import capi
        ctx.set_detail_mode(capi.SK_DETAIL_MATCHED)
        pad = 300
        d_bc = ctx.malloc_device(bc.nbytes + 64)
        outs = [(np.int32, 4), (np.uint8, 1), (np.int16, 2), (np.int16, 2)]
        d_out = [ctx.malloc_device((n + pad) * w) for _, w in outs]
        try:
            ctx.copy_h2d(d_bc, bc)
            guards = [np.full(n + pad, 0x5A5A5A5A if w == 4 else (0x5A if w == 1 else 0x5A5A), dtype=t) for t, w in outs]
            for p, g in zip(d_out, guards):
                ctx.copy_h2d(p, g)
            ctx.demux_assign_dev(d_bc, bc.shape[1], n, *d_out)
            ctx.sync()
            got = [np.empty(n + pad, dtype=t) for t, _ in outs]
            for p, g in zip(d_out, got):
                ctx.copy_d2h(g, p)
        finally:
            ctx.free_device(d_bc)
            for p in d_out:
                ctx.free_device(p)
            ctx.set_detail_mode(capi.SK_DETAIL_FULL)
        e = oracle.demux_batch(table, bc, 1)
        assert np.array_equal(got[0][:n], e[0])
        m = e[0] != -1
        assert np.array_equal(got[1][:n][m], e[1][m]) and np.array_equal(got[2][:n][m], e[2][m]) and np.array_equal(got[3][:n][m], e[3][m])
        for g, gd in zip(got, guards):
            assert np.array_equal(g[n:], gd[n:]), "a store behind the last row"


def test_demux_by_table_rows_with_wildcards(ctx, oracle, lut_form):
    """A sheet row may hold `N` / `U` where other rows hold a letter (src/fasta_demultiplex.rs:272: such a column does not count
    for THAT row).  The table's builder enumerates the row once per class of the column (sk_lut.cpp), so these sheets take the
    lookup as well: SURVEY.md Appendix A's pair, and 40 samples with up to two wildcards per row."""
    table = np.frombuffer(b"ACGTACGTACGTNCGTTTGCAANNNGGATCCACATGCATG", dtype=np.uint8).reshape(5, 8).copy()
    rng = np.random.default_rng(5)
    bc, _ = synth.observe_barcodes(np.where(table == ord("N"), ord("A"), table).astype(np.uint8), 60_001, seed=5)
    bc[::7, 4] = ord("N")
    bc[::11, 6:] = rng.choice(synth.BASES, size=(bc[::11].shape[0], 2))
    check_demux_decision_only(ctx, oracle, table, bc)
    assert check_demux_matched(ctx, oracle, table, bc) > 20_000
    check_demux_matched(ctx, oracle, table, bc, max_diff=0)
    table = synth.make_sheet(40, 8, dual=True, seed=40)
    clean = table.copy()
    for s in range(0, 40, 3):
        table[s, rng.integers(0, 8)] = ord("N")
        if s % 2 == 0:
            table[s, 9 + rng.integers(0, 8)] = ord("U")
    bc, _ = synth.observe_barcodes(clean, 80_001, seed=41, halves=2)
    bc[::13, 2] = ord("N")
    check_demux_decision_only(ctx, oracle, table, bc)
    assert check_demux_matched(ctx, oracle, table, bc) > 40_000


def test_fused_pass_of_a_large_sheet_takes_the_table(ctx, oracle):
    """384 samples are beyond the tile pass's own matcher (128): the barcode phase of a fused call is then a launch of its own
    and takes the lookup like a demultiplex-alone call — same answers as the oracle's three commands."""
    n, L = 20_011, 150
    table = synth.make_sheet(384, 8, dual=True, seed=384)
    bc, _ = synth.observe_barcodes(table, n, seed=5, halves=2)
    seq, qual = synth.make_reads(n, L, seed=70)
    qual = synth.add_forced_classes(qual, seed=80)
    ctx.set_barcodes(table, 1)
    r = ctx.fused_pass([(seq, qual, None)], 20, bc=bc, want_detail=False)
    e = oracle.demux_batch(table, bc, 1)
    assert np.array_equal(r["assign"], e[0]) and np.array_equal(ctx.counts(), e[4])
    assert np.array_equal(r["lowest_k"][0], oracle.trim_batch(qual, None, 20))
    assert np.array_equal(r["out_seq"][0], oracle.mask_batch(seq, qual, None, 20))


@pytest.mark.parametrize("S", [96, 384])
def test_fused_pass_of_a_mixed_case_sheet(ctx, oracle, S):
    """A sheet typed partly in lower case has more letters than the tile pass's own matcher knows (seven): the barcode phase of a
    fused call is a launch of its own, served by the table with wide classes (sk_lut.h) — the oracle's three commands' answers,
    with and without the detail columns of matched rows."""
    from seqkit_amd import capi
    n, L = 30_011, 150
    table = synth.make_sheet(S, 8, dual=True, seed=S + 1)
    table[1::2] = np.where((table[1::2] >= 65) & (table[1::2] <= 90), table[1::2] + 32, table[1::2]).astype(np.uint8)
    bc, _ = synth.observe_barcodes(table, n, seed=6, halves=2)
    bc[::17, 3] = ord("N")
    seq, qual = synth.make_reads(n, L, seed=71)
    qual = synth.add_forced_classes(qual, seed=81)
    ctx.set_barcodes(table, 1)
    assert ctx.barcode_table_info()["kind"] == capi.SK_TABLE_FACTORED | capi.SK_TABLE_WIDE_CLASSES
    e = oracle.demux_batch(table, bc, 1)
    for detail_mode, want_detail in ((capi.SK_DETAIL_FULL, False), (capi.SK_DETAIL_MATCHED, True), (capi.SK_DETAIL_FULL, True)):
        ctx.set_detail_mode(detail_mode)
        ctx.counts_reset()
        r = ctx.fused_pass([(seq, qual, None)], 20, bc=bc, want_detail=want_detail)
        assert np.array_equal(r["assign"], e[0]) and np.array_equal(ctx.counts(), e[4])
        if want_detail:
            m = e[0] != -1 if detail_mode == capi.SK_DETAIL_MATCHED else np.ones(n, dtype=bool)
            assert np.array_equal(r["lowest_diff"][m], e[1][m]) and np.array_equal(r["first_idx"][m], e[2][m]) and np.array_equal(r["last_idx"][m], e[3][m])
        assert np.array_equal(r["lowest_k"][0], oracle.trim_batch(qual, None, 20))
        assert np.array_equal(r["out_seq"][0], oracle.mask_batch(seq, qual, None, 20))
    ctx.set_detail_mode(capi.SK_DETAIL_FULL)


def test_demux_by_table_ambiguity_duplicates_umi(ctx, oracle, lut_form):
    """Ambiguous keys keep their (first, last) pair in the side list; duplicates are always ambiguous; UMI columns do not count."""
    table = np.array([list(b"ACGTACGTAAAA"), list(b"ACGTACGAAAAT"), list(b"TTCTTTTTUUUU"), list(b"GGGGGGGGUUUU"),
                      list(b"CCCCCCCCCCCC"), list(b"CCCCCCCCCCCC")], dtype=np.uint8)
    table[:, 8:] = np.frombuffer(b"UUUU", dtype=np.uint8)
    rng = np.random.default_rng(9)
    src = table[rng.integers(0, 6, size=50000)].copy()
    alphabet = np.frombuffer(b"ACGTNU+acgt", dtype=np.uint8)
    for _ in range(2):
        hit = rng.random(src.shape[0]) < 0.5
        pos = rng.integers(0, 12, size=src.shape[0])
        src[hit, pos[hit]] = alphabet[rng.integers(0, alphabet.size, size=int(hit.sum()))]
    check_demux_decision_only(ctx, oracle, table, src)
    check_demux_matched(ctx, oracle, table, src)
    e = oracle.demux_batch(table, src, 1)
    assert (e[0] == -2).sum() > 1000
    table = np.array([list(b"AAAAAAAA"), list(b"AAAAAATT"), list(b"AAAAAATA"), list(b"TTTTTTTT")], dtype=np.uint8)      # Appendix A's ambiguity, and a chain of neighbours
    src = table[rng.integers(0, 4, size=20000)].copy()
    hit = rng.random(20000) < 0.7
    src[hit, rng.integers(0, 8, size=20000)[hit]] = synth.BASES[rng.integers(0, 4, size=int(hit.sum()))]
    check_demux_matched(ctx, oracle, table, src)


@pytest.mark.parametrize("chunk_log2", [None, 17])
@pytest.mark.parametrize("want_detail", [False, True])
def test_demux_counters_across_launches(ctx, oracle, monkeypatch, want_detail, chunk_log2):
    """The counters of demultiplex alone go through per-launch copies and tickets that the last workgroup of a launch leaves
    zero: one sheet, many launches of every grid size (one workgroup, fewer workgroups than copies, a full grid), counters
    accumulated over all of them and compared once per launch; with small chunks a call is many launches on the two lanes of
    the host pipeline, in flight together, each lane with its own set of copies."""
    if chunk_log2:
        monkeypatch.setenv("SK_HOST_CHUNK_LOG2", str(chunk_log2))
    for S, dual in ((16, False), (96, True)):
        table = synth.make_sheet(S, 8, dual=dual, seed=11)
        bc, _ = synth.observe_barcodes(table, 700_001, seed=12, halves=2 if dual else 1)
        ctx.set_barcodes(table, 1)
        e_assign, _, _, _, e_counts = oracle.demux_batch(table, bc, 1)

        def counts_of(a):          # the oracle's counters of a prefix, from its per-row decisions: per sample, total, identified, ambiguous
            return np.concatenate([np.bincount(a[a >= 0], minlength=S), [a.size, (a >= 0).sum(), (a == -2).sum()]]).astype(np.uint64)
        assert np.array_equal(counts_of(e_assign), e_counts.astype(np.uint64))
        expect = np.zeros(S + 3, dtype=np.uint64)
        for n in (1, 63, 64, 65, 1000, 256 * 9, 256 * 17 + 5, 100_000, 700_001, 64 * 4 * 15, 3, 300_000):
            assign, *_ = ctx.demux_assign(bc[:n], want_detail=want_detail)
            assert np.array_equal(assign, e_assign[:n])
            expect += counts_of(e_assign[:n])
            assert np.array_equal(ctx.counts().astype(np.uint64), expect), (S, n)
        ctx.counts_reset()
        assign, *_ = ctx.demux_assign(bc[:5000], want_detail=want_detail)
        assert np.array_equal(ctx.counts().astype(np.uint64), counts_of(e_assign[:5000]))


def test_matcher_and_table_launches_share_the_ctx_counters(ctx, oracle):
    """A call with the full detail columns runs the matchers, a call without them the table lookup; both add to the ctx's
    counters, the lookups through the wide copies whose `identified` the fold derives from the samples' bins (sk_kernels.hip,
    counts_fold_wide_kernel).  Interleaved on one sheet: the counters are the oracle's, and identified == sum of the samples."""
    for S, dual in ((16, False), (96, True), (200, True)):
        table = synth.make_sheet(S, 8, dual=dual, seed=21)
        bc, _ = synth.observe_barcodes(table, 300_000, seed=22, halves=2 if dual else 1)
        ctx.set_barcodes(table, 1)
        ctx.counts_reset()
        e = oracle.demux_batch(table, bc, 1)
        for k, want_detail in enumerate((True, False, False, True, False)):
            assign, *_ = ctx.demux_assign(bc, want_detail=want_detail)
            assert np.array_equal(assign, e[0])
            got = ctx.counts().astype(np.uint64)
            assert np.array_equal(got, (k + 1) * e[4].astype(np.uint64)), (S, k, got[S:], e[4][S:])
            assert int(got[S + 1]) == int(got[:S].sum())


@pytest.mark.parametrize("seed", range(int(os.environ.get("SK_FUZZ_SEEDS", "48"))))      # SK_FUZZ_SEEDS=400: the long run, once per round on the GPU box
def test_fuzz_demux_by_table(ctx, oracle, seed, monkeypatch):
    """Sheets with and without a lookup table: wildcard columns (all rows / some rows), a separator, duplicates, lower case
    next to upper case, up to 7 letters and more, any length to 20 and above, more than 128 samples, max_diff 0 / 1 / 2,
    observed barcodes with foreign bytes, padded rows; every form of the lookup kernel; decision only, matched detail, and
    the full detail right after on the same sheet."""
    if seed % 4 == 1:
        monkeypatch.setenv("SK_DEMUX_LDSTAB", "0")          # the table from the vector cache instead of LDS
    if seed % 4 == 2:
        monkeypatch.setenv("SK_DEMUX_DIRECT", "0")          # aligned short rows by the gather loads too
    if seed % 4 == 3:
        monkeypatch.setenv("SK_DEMUX_ROWS2", "1")           # 8-byte rows two per lane also for the decision alone
    rng = np.random.default_rng(12000 + seed)
    S = int(rng.choice([1, 2, 3, 16, 40, 96, 128, 150, 129, 384, 1000]))
    L = int(rng.choice([1, 3, 4, 8, 9, 12, 16, 17, 20, 21, 24, 33]))
    alphabet = [b"ACGT", b"ACGTN", b"ACGT+", b"ACGTacgt", b"ACGTRYKM", b"AC", b"ACGTN+U", b"ACGT-_x"][int(rng.integers(0, 8))]
    table = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=(S, L)).astype(np.uint8)
    kind = int(rng.integers(0, 5))
    if seed % 6 == 5:                             # a dual-index sheet out of two sets of half-barcodes (the half-by-half form when they are far apart)
        S = int(rng.choice([150, 384, 1000]))
        hl = int(rng.choice([4, 8]))
        n7 = int(rng.integers(8, 40)) if hl == 8 else int(rng.integers(2, 8))
        i7 = synth._distant_kmers(n7, hl, int(rng.choice([1, 3])), rng)
        i5 = synth._distant_kmers(min((S + n7 - 1) // n7 + 1, 130 if hl == 8 else 10), hl, 3, rng)      # (at most 16 4-mers are 3 apart)
        S = min(S, n7 * i5.shape[0])
        combo = np.where(rng.random(S) < 0.1, rng.integers(0, n7 * i5.shape[0], size=S), np.arange(S))      # mostly distinct combinations, some twice
        table = np.concatenate([i7[combo % n7], np.full((S, 1), ord("+"), dtype=np.uint8), i5[combo // n7]], axis=1).astype(np.uint8)
        L, kind = 2 * hl + 1, 0
    if kind == 1 and L >= 4:                     # UMI columns: a wildcard in every row
        table[:, L - 3:] = ord("U")
    elif kind == 2:                               # wildcards here and there (no table then)
        table[rng.random((S, L)) < 0.1] = ord("N")
    elif kind >= 3 and L >= 3:                    # a separator column
        table[:, int(rng.integers(0, L))] = ord("+") if kind == 3 else ord("A")
    if S >= 3 and rng.random() < 0.3:
        table[2] = table[0]                       # duplicate barcodes: always ambiguous
    n = int(rng.choice([1, 64, 777, 5000, 70001]))
    pick = table[rng.integers(0, S, size=n)].copy()
    noise = rng.random((n, L)) < 0.05
    pick[noise] = rng.choice(np.frombuffer(b"ACGTNacgtn+U\x00\xff#", dtype=np.uint8), size=int(noise.sum()))
    stride = L + int(rng.choice([0, 0, 1, 3, 7])) if L <= 24 else L
    bc = np.full((n, stride), 0x43, dtype=np.uint8)
    bc[:, :L] = pick
    table = np.ascontiguousarray(table)
    md = int(rng.choice([0, 1, 1, 1, 2]))
    e = oracle.demux_batch(table, np.ascontiguousarray(pick), md)
    ctx.set_barcodes(table, md)
    assign, low, *_ = ctx.demux_assign(bc, want_detail=False)
    assert low is None and np.array_equal(assign, e[0]) and np.array_equal(ctx.counts(), e[4])
    import seqkit_amd
    ctx.counts_reset()
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
    try:
        assign, low, first, last = ctx.demux_assign(bc)
    finally:
        ctx.set_detail_mode(seqkit_amd.SK_DETAIL_FULL)
    m = e[0] != -1
    assert np.array_equal(assign, e[0]) and np.array_equal(ctx.counts(), e[4])
    assert np.array_equal(low[m], e[1][m]) and np.array_equal(first[m], e[2][m]) and np.array_equal(last[m], e[3][m])
    assign, low, first, last = ctx.demux_assign(bc)           # and the full detail (the matchers) on the same sheet
    assert np.array_equal(assign, e[0]) and np.array_equal(low, e[1]) and np.array_equal(first, e[2]) and np.array_equal(last, e[3])


# ---- fused pass ---------------------------------------------------------------------------------------------
def test_tile_pass_shapes(ctx, oracle):
    for n, L, paired in ((30011, 150, True), (777, 150, False), (5003, 101, True), (64 * 300 + 1, 33, True)):
        table = synth.make_sheet(96, 8, dual=True, seed=4)
        bc, _ = synth.observe_barcodes(table, n, seed=n, halves=2)
        ln = synth.ragged_lengths(n, L, seed=n) if n == 5003 else None
        mates = []
        for mi in range(2 if paired else 1):
            seq, qual = synth.make_reads(n, L, seed=70 + mi)
            mates.append((seq, synth.add_forced_classes(qual, seed=80 + mi), ln))
        ctx.set_barcodes(table, 1)
        r = ctx.fused_pass(mates, 20, bc=bc, want_detail=True)
        e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, 1)
        assert np.array_equal(r["assign"], e_assign) and np.array_equal(r["lowest_diff"], e_low)
        assert np.array_equal(r["first_idx"], e_first) and np.array_equal(r["last_idx"], e_last)
        assert np.array_equal(ctx.counts(), e_counts)
        for mi, (seq, qual, _) in enumerate(mates):
            assert np.array_equal(r["lowest_k"][mi], oracle.trim_batch(qual, ln, 20))
            exp = oracle.mask_batch(seq, qual, ln, 20)
            if ln is None:
                assert np.array_equal(r["out_seq"][mi], exp)
            else:
                for i in range(0, n, 7):
                    assert np.array_equal(r["out_seq"][mi][i, :ln[i]], exp[i, :ln[i]])
        # trim alone and mask+trim without barcodes go through the same kernels
        assert np.array_equal(ctx.trim_by_quality(mates[0][1], ln, 30), oracle.trim_batch(mates[0][1], ln, 30))
        r2 = ctx.fused_pass(mates, 2)
        assert np.array_equal(r2["lowest_k"][-1], oracle.trim_batch(mates[-1][1], ln, 2))



@pytest.mark.parametrize("paired", [False, True])
def test_fused_pass_matches_the_three_commands(ctx, oracle, paired):
    """cfg 4 shape: add barcode + demultiplex + trim + mask in one pass == the separate oracle steps."""
    n, L = 30011, 150
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=21, halves=2)
    mates = []
    for mi in range(2 if paired else 1):
        seq, qual = synth.make_reads(n, L, seed=30 + mi)
        qual = synth.add_forced_classes(qual, seed=40 + mi)
        mates.append((seq, qual, None))
    ctx.set_barcodes(table, 1)
    r = ctx.fused_pass(mates, 20, bc=bc, want_detail=True)
    e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, 1)
    assert np.array_equal(r["assign"], e_assign) and np.array_equal(r["lowest_diff"], e_low)
    assert np.array_equal(r["first_idx"], e_first) and np.array_equal(r["last_idx"], e_last)
    assert np.array_equal(ctx.counts(), e_counts)
    for mi, (seq, qual, _) in enumerate(mates):
        assert np.array_equal(r["lowest_k"][mi], oracle.trim_batch(qual, None, 20))
        assert np.array_equal(r["out_seq"][mi], oracle.mask_batch(seq, qual, None, 20))


def test_fused_pass_ragged(ctx, oracle):
    n, L = 5003, 101
    seq, qual = synth.make_reads(n, L, seed=50)
    ln = synth.ragged_lengths(n, L, seed=50)
    table = synth.make_sheet(16, 8, dual=False, seed=3)
    bc, _ = synth.observe_barcodes(table, n, seed=51)
    ctx.set_barcodes(table, 1)
    r = ctx.fused_pass([(seq, qual, ln)], 25, bc=bc)
    assert np.array_equal(r["assign"], oracle.demux_batch(table, bc, 1)[0])
    assert np.array_equal(r["lowest_k"][0], oracle.trim_batch(qual, ln, 25))
    exp = oracle.mask_batch(seq, qual, ln, 25)
    for i in range(n):          # bytes past len[r] are unspecified on output
        assert np.array_equal(r["out_seq"][0][i, :ln[i]], exp[i, :ln[i]])


def test_fused_pass_multichunk_host_staging(ctx, oracle):
    """More rows than one staging chunk of the host entry point (192 MiB of workspace per chunk)."""
    n, L = 700000, 150
    rng = np.random.default_rng(60)
    qual = rng.integers(33, 75, size=(n, L), dtype=np.uint8)
    seq = synth.BASES[rng.integers(0, 4, size=(n, L))]
    r = ctx.fused_pass([(seq, qual, None)], 20)
    assert np.array_equal(r["lowest_k"][0], oracle.trim_batch(qual, None, 20))
    assert np.array_equal(r["out_seq"][0], oracle.mask_batch(seq, qual, None, 20))


# ---- the same pass over the tile-blocked batch layout (sk_fused_pass_blocked_dev) ---------------------------------
def check_blocked(ctx, oracle, mates, m, table=None, bc=None, md=1, do_mask=True, do_trim=True, detail=True):
    n, stride = mates[0][1].shape
    if table is not None:
        ctx.set_barcodes(table, md)
    r = ctx.fused_pass_blocked(mates, m, bc=bc, want_detail=detail, do_mask=do_mask, do_trim=do_trim)
    if bc is not None:
        e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, md)
        assert np.array_equal(r["assign"], e_assign)
        if detail:
            assert np.array_equal(r["lowest_diff"], e_low)
            assert np.array_equal(r["first_idx"], e_first) and np.array_equal(r["last_idx"], e_last)
        assert np.array_equal(ctx.counts(), e_counts)
    for i, (seq, qual, ln) in enumerate(mates):
        if do_trim:
            assert np.array_equal(r["lowest_k"][i], oracle.trim_batch(qual, ln, m)), i
        if do_mask:
            exp = oracle.mask_batch(seq, qual, ln, m)
            valid = np.ones((n, stride), dtype=bool) if ln is None else (np.arange(stride)[None, :] < ln[:, None])
            assert np.array_equal(r["out_seq"][i][valid], exp[valid]), i


@pytest.mark.parametrize("n,L,paired,ragged", [(30011, 150, True, False), (777, 150, False, False), (5003, 101, True, True),
                                               (64 * 300, 33, True, False), (1, 150, True, False), (63, 250, False, True),
                                               (4097, 151, True, False), (130, 960, False, False)])
def test_blocked_pass_cfg4_shapes(ctx, oracle, n, L, paired, ragged):
    """cfg 4 (paired 2x150, 96 dual-index barcodes) and neighbours in the tile-blocked layout == the three oracle steps."""
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=n, halves=2)
    ln = synth.ragged_lengths(n, L, seed=n) if ragged else None
    mates = []
    for mi in range(2 if paired else 1):
        seq, qual = synth.make_reads(n, L, seed=170 + mi)
        mates.append((seq, synth.add_forced_classes(qual, seed=180 + mi), ln))
    check_blocked(ctx, oracle, mates, 20, table, bc)
    check_blocked(ctx, oracle, mates, 20, table, bc, detail=False)
    check_blocked(ctx, oracle, mates, 30, do_mask=False)                 # trim alone, no barcodes
    check_blocked(ctx, oracle, mates, 2, do_trim=False)                  # mask alone
    check_blocked(ctx, oracle, mates, 41, table, bc, do_mask=False)      # demultiplex + trim


@pytest.mark.parametrize("min_baseq", [0, 1, 20, 95, 96, 222, 223, 224, 255])
def test_blocked_pass_all_threshold_modes(ctx, oracle, min_baseq):
    rng = np.random.default_rng(900 + min_baseq)
    n, stride = 2000, 150
    mates = []
    for _ in range(2):
        seq = synth.BASES[rng.integers(0, 4, size=(n, stride))]
        qual = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
        mates.append((seq, qual, None))
    check_blocked(ctx, oracle, mates, min_baseq)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_blocked_pass_random_shapes(ctx, oracle, seed):
    """Random rows / stride / lengths / mates / mask-trim mix / sheets the blocked pass serves / thresholds / bytes."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 63, 64, 65, 200, 1023, 4100]))
    stride = int(rng.choice([1, 2, 7, 33, 64, 100, 150, 151, 255, 640, 960]))
    if stride > 600:
        n = min(n, 300)
    nm = int(rng.integers(1, 3))
    do_mask, do_trim = [(True, True), (True, False), (False, True)][int(rng.integers(0, 3))]
    m = int(rng.choice([0, 1, 2, 20, 30, 41, 95, 200, 223, 224, 255]))
    ragged = rng.random() < 0.5
    ln = synth.ragged_lengths(n, stride, seed=seed) if ragged else None
    mates = []
    for _ in range(nm):
        if rng.random() < 0.4:
            seq = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
            qual = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
        else:
            seq, qual = synth.make_reads(n, stride, seed=int(rng.integers(0, 1 << 30)))
            qual = synth.add_forced_classes(qual, seed=seed)
        mates.append((seq, qual, ln))
    bc = table = None
    md = 1
    if rng.random() < 0.7:
        S = int(rng.choice([1, 2, 16, 96, 128]))
        L = int(rng.choice([4, 8, 17, 24, 31]))
        table = np.ascontiguousarray(rng.choice(np.frombuffer(b"ACGTNU", dtype=np.uint8), size=(S, L), p=[.23, .23, .23, .23, .05, .03]))
        pick = table[rng.integers(0, S, size=n)].copy()
        noise = rng.random((n, L)) < 0.08
        pick[noise] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(noise.sum()))
        bc = np.ascontiguousarray(pick)
        md = int(rng.choice([0, 1, 1, 2, 5]))
    check_blocked(ctx, oracle, mates, m, table, bc, md, do_mask, do_trim)


def test_blocked_pass_rejects_what_it_does_not_serve(ctx):
    """Shapes outside the tile-blocked pass fail loudly (SK_ERR_INVALID) instead of taking another path silently."""
    import seqkit_amd
    from seqkit_amd import capi
    n = 100
    seq, qual = synth.make_reads(n, 1100, seed=1)
    with pytest.raises(seqkit_amd.SeqkitHipError):
        ctx.fused_pass_blocked([(seq, qual, None)], 20)                  # rows longer than an LDS tile
    table = np.frombuffer(b"ACGTRYKM" * 4, dtype=np.uint8).reshape(4, 8).copy()   # 8 distinct bytes: no bit-sliced tables
    table[1] = np.frombuffer(b"WSBDHVXZ", dtype=np.uint8)
    ctx.set_barcodes(table, 1)
    seq, qual = synth.make_reads(n, 50, seed=1)
    with pytest.raises(seqkit_amd.SeqkitHipError):
        ctx.fused_pass_blocked([(seq, qual, None)], 20, bc=np.ascontiguousarray(table[np.zeros(n, dtype=int)]))
    with pytest.raises(seqkit_amd.SeqkitHipError):
        capi.blocked_layout(0, 150, 17, 3)


# ---- BAM ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,max_frag", [(1, 5000), (1000, 5000), (300001, 5000), (50000, 100), (50000, 0), (20000, 20000)])
def test_bam_flag_tlen(ctx, oracle, n, max_frag):
    flag, tid, mtid, tlen = synth.make_bam_cores(n, seed=5)
    if n > 10:
        tlen[3] = -2147483648
        tlen[4] = 2147483647
        flag[3] = flag[4] = 0x1 | 0x40
        mtid[3], mtid[4] = tid[3], tid[4]
    c, h, t = ctx.bam_flag_tlen(flag, tid, mtid, tlen, max_frag)
    ec, eh, et = oracle.bam_flag_tlen(flag, tid, mtid, tlen, max_frag)
    assert np.array_equal(c, ec) and np.array_equal(h, eh) and t == et
    assert int(h.sum()) == t


def test_bam_dev_entry_point_aligned_and_unaligned_columns(ctx, oracle):
    """_dev entry point (device memory through the C-ABI's own helpers): 16-byte aligned columns take the vectorised
    kernel, offset columns the scalar one."""
    n = 100003
    flag, tid, mtid, tlen = synth.make_bam_cores(n + 8, seed=6)
    cols = [flag, tid, mtid, tlen]
    dptr = [ctx.malloc_device(c.nbytes) for c in cols]
    dout = ctx.malloc_device((4 + 5001) * 8)
    try:
        for c, d in zip(cols, dptr):
            ctx.copy_h2d(d, c)
        for off in (0, 3):
            zero = np.zeros(4 + 5001, dtype=np.uint64)
            ctx.copy_h2d(dout, zero)
            ctx.bam_flag_tlen_dev(dptr[0] + 2 * off, dptr[1] + 4 * off, dptr[2] + 4 * off, dptr[3] + 4 * off, n, 5000, dout)
            got = np.empty(4 + 5001, dtype=np.uint64)
            ctx.copy_d2h(got, dout)
            ctx.sync()
            ec, eh, et = oracle.bam_flag_tlen(flag[off:off + n], tid[off:off + n], mtid[off:off + n], tlen[off:off + n], 5000)
            assert np.array_equal(got[:3], ec) and int(got[3]) == et and np.array_equal(got[4:], eh), off
    finally:
        for d in dptr + [dout]:
            ctx.free_device(d)


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 5000), (100003, 0, 5000), (65536, 100, 300), (5000, -5, 10**12), (5000, 400, 100), (777, 0, 0)])
def test_bam_fragments_filter(ctx, oracle, n, lo, hi):
    """f2, src/sam_fragments.rs:27-38: the keep mask and the count against the oracle, incl. empty and unbounded windows."""
    flag, tid, mtid, tlen = synth.make_bam_cores(n, seed=n)
    flag ^= (np.random.default_rng(n).random(n) < 0.05).astype(np.uint16) * 0x200     # some QC failures
    if n > 10:
        tlen[5] = -2147483648
        flag[5] = 0x1 | 0x20
        mtid[5] = tid[5]
    keep, kept = ctx.bam_fragments(flag, tid, mtid, tlen, lo, hi)
    exp = oracle.fragments_keep(flag, tid, mtid, tlen, lo, hi)
    assert np.array_equal(keep, exp) and kept == int(exp.sum())


# ---- error behaviour of the boundary ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(160))
def test_fuzz_fused_pass_random_shapes(ctx, oracle, seed):
    """Random everything: rows, stride (LDS tile path and the long-row path), ragged or full lengths, one or two mates, mask
    and/or trim, with or without barcodes, any sheet size / barcode length / mismatch budget / threshold, arbitrary bytes."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 63, 64, 65, 200, 1023, 4100]))
    stride = int(rng.choice([1, 2, 7, 33, 64, 100, 150, 151, 255, 640, 1100]))
    if stride > 600:
        n = min(n, 300)
    nm = int(rng.integers(1, 3))
    do_mask, do_trim = [(True, True), (True, False), (False, True)][int(rng.integers(0, 3))]
    m = int(rng.choice([0, 1, 2, 20, 30, 41, 95, 200, 223, 224, 255]))
    arbitrary = rng.random() < 0.4
    mates = []
    for _ in range(nm):
        if arbitrary:
            seq = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
            qual = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
        else:
            seq, qual = synth.make_reads(n, stride, seed=int(rng.integers(0, 1 << 30)))
            qual = synth.add_forced_classes(qual, seed=seed)
        ln = synth.ragged_lengths(n, stride, seed=seed) if rng.random() < 0.5 else None
        mates.append((seq, qual, ln))
    bc = table = None
    md = 1
    if rng.random() < 0.7:
        S = int(rng.choice([1, 2, 16, 96, 130, 300]))
        L = int(rng.choice([4, 8, 17, 24, 33]))
        table = rng.choice(np.frombuffer(b"ACGTNU", dtype=np.uint8), size=(S, L), p=[.23, .23, .23, .23, .05, .03])
        pick = table[rng.integers(0, S, size=n)].copy()
        noise = rng.random((n, L)) < 0.08
        pick[noise] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(noise.sum()))
        bc = np.ascontiguousarray(pick)
        md = int(rng.choice([0, 1, 1, 2, 5]))
        ctx.set_barcodes(np.ascontiguousarray(table), md)
    r = ctx.fused_pass(mates, m, bc=bc, want_detail=True, do_mask=do_mask, do_trim=do_trim)
    if bc is not None:
        e_assign, e_low, e_first, e_last, _ = oracle.demux_batch(table, bc, md)
        assert np.array_equal(r["assign"], e_assign) and np.array_equal(r["lowest_diff"], e_low)
        assert np.array_equal(r["first_idx"], e_first) and np.array_equal(r["last_idx"], e_last)
    for i, (seq, qual, ln) in enumerate(mates):
        if do_trim:
            assert np.array_equal(r["lowest_k"][i], oracle.trim_batch(qual, ln, m)), (seed, i)
        if do_mask:
            exp = oracle.mask_batch(seq, qual, ln, m)
            valid = np.ones((n, stride), dtype=bool) if ln is None else (np.arange(stride)[None, :] < ln[:, None])
            assert np.array_equal(r["out_seq"][i][valid], exp[valid]), (seed, i)


# ---- fasta gc content ------------------------------------------------------------------------------------------------
def test_gc_count_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(17)
    genome = rng.choice(np.frombuffer(b"ACGTNacgtnRYKM-*", dtype=np.uint8), size=3_000_000, p=[.2, .2, .2, .2, .04, .03, .03, .03, .03, .01] + [.005] * 6)
    genome = np.ascontiguousarray(genome)
    ctx.gc_set_genome(genome)
    starts = np.concatenate([rng.integers(0, len(genome) - 200_000, size=400), [0, 0, len(genome), len(genome) - 1, 5]]).astype(np.int64)
    lens = np.concatenate([rng.choice([0, 1, 15, 16, 17, 100, 1000, 65535, 65536, 65537, 200_000], size=400), [len(genome), 0, 0, 1, 31]]).astype(np.int64)
    gc, total = ctx.gc_count(starts, lens)
    raw = genome.tobytes()
    for i in range(len(starts)):
        assert (int(gc[i]), int(total[i])) == oracle.gc_count(raw[starts[i]:starts[i] + lens[i]]), (i, starts[i], lens[i])
    assert ctx.gc_count(np.zeros(0, np.int64), np.zeros(0, np.int64))[0].size == 0
    from seqkit_amd.capi import SeqkitHipError
    with pytest.raises(SeqkitHipError):
        ctx.gc_count(np.array([len(genome) - 5]), np.array([6]))                # leaves the genome
    ctx.gc_set_genome(b"")                                                       # an empty genome is a genome
    assert [int(x) for x in ctx.gc_count(np.array([0]), np.array([0]))[0]] == [0]


# ---- f2 (second half): sam count ---------------------------------------------------------------------------------
def count_inputs(n, n_chr, n_regions, seed, span=2_000_000):
    """Coordinate-sorted record columns and BED-like regions (unsorted, overlapping, nested, empty, some on a chromosome
    the BAM does not have)."""
    rng = np.random.default_rng(seed)
    tid = np.sort(rng.integers(0, n_chr, size=n)).astype(np.int32)
    pos = rng.integers(0, span, size=n).astype(np.int32)
    order = np.lexsort((pos, tid))
    tid, pos = tid[order], pos[order]
    flag = rng.choice(np.array([99, 147, 83, 163, 65, 129, 97, 145, 0, 16, 4, 1024 + 99, 256 + 99, 2048 + 99, 73, 1], dtype=np.uint16), size=n)
    mapq = rng.integers(0, 61, size=n).astype(np.uint8)
    tlen = rng.integers(-700, 700, size=n).astype(np.int32)
    odd = np.nonzero(rng.random(n) < 0.02)[0]
    tlen[odd] = rng.choice(np.array([0, 19, 20, -20, 5000, 5001, -2**31, 2**31 - 1], dtype=np.int64), size=len(odd)).astype(np.int32)
    mpos = (pos.astype(np.int64) + tlen.astype(np.int64) // 2).astype(np.int32)
    same = rng.random(n) < 0.1
    mpos[same] = pos[same]
    mtid = tid.copy()
    mtid[rng.random(n) < 0.05] = -1
    end_pos = (pos + rng.integers(0, 160, size=n)).astype(np.int32)
    rchr = rng.integers(-1, n_chr, size=n_regions).astype(np.int32)
    rstart = rng.integers(0, span, size=n_regions).astype(np.uint32)
    rlen = rng.choice(np.array([0, 1, 50, 300, 2000, 100000], dtype=np.uint32), size=n_regions)
    rend = (rstart + rlen).astype(np.uint32)
    return dict(flag=flag, mapq=mapq, tid=tid, mtid=mtid, pos=pos, mpos=mpos, tlen=tlen, end_pos=end_pos), rchr, rstart, rend


def grouped_regions(rchr, rstart, rend, n_chr):
    """The C-ABI takes regions grouped by reference (any order inside a group) + their original indices."""
    keep = np.nonzero(rchr >= 0)[0]
    order = keep[np.argsort(rchr[keep], kind="stable")]
    chr_off = np.zeros(n_chr + 1, dtype=np.int32)
    np.add.at(chr_off, rchr[keep] + 1, 1)
    return np.cumsum(chr_off).astype(np.int32), rstart[order], rend[order], order.astype(np.int32)


@pytest.mark.parametrize("kw", [dict(), dict(min_mapq=30), dict(max_frag_len=300), dict(single_end=True), dict(center=True),
                                dict(single_end=True, center=True, min_mapq=10, max_frag_len=100), dict(max_frag_len=0), dict(max_frag_len=2**32 - 1)])
def test_sam_count_matches_oracle(ctx, oracle, kw):
    n_chr = 5
    cols, rchr, rstart, rend = count_inputs(200_000, n_chr, 3000, seed=9)
    want, code, _ = oracle.count_batch(**cols, n_chr=n_chr, rchr=rchr, rstart=rstart, rend=rend, **kw)
    assert code == 0 and (want.sum() > 1000 or kw.get("max_frag_len") == 0)
    chr_off, gs, ge, gi = grouped_regions(rchr, rstart, rend, n_chr)
    ctx.count_set_regions(chr_off, gs, ge, gi, n_regions=3000)
    got_cols = dict(cols)
    if not kw.get("single_end"):
        got_cols["end_pos"] = None
    half = 77_777                                                    # two batches: counters accumulate
    ctx.count_add(**{k: (None if v is None else v[:half]) for k, v in got_cols.items()}, **kw)
    ctx.count_add(**{k: (None if v is None else v[half:]) for k, v in got_cols.items()}, **kw)
    got = ctx.count_get()
    assert len(got) == 3000 and np.array_equal(got[gi], want[gi]) and got[rchr < 0].sum() == 0
    ctx.count_set_regions(chr_off, gs, ge, gi, n_regions=3000)       # setting the regions again clears the counters
    assert ctx.count_get().sum() == 0


def test_sam_count_edge_cases(ctx, oracle):
    # no regions at all, one chromosome without regions, a region list with one entry
    cols, _, _, _ = count_inputs(1000, 2, 1, seed=3)
    ctx.count_set_regions(np.array([0, 0, 0], dtype=np.int32), np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    ctx.count_add(**dict(cols, end_pos=None))
    assert len(ctx.count_get()) == 0
    rchr, rstart, rend = np.array([1], np.int32), np.array([0], np.uint32), np.array([2_000_000], np.uint32)
    want, code, _ = oracle.count_batch(**cols, n_chr=2, rchr=rchr, rstart=rstart, rend=rend)
    ctx.count_set_regions(np.array([0, 0, 1], dtype=np.int32), rstart, rend)
    ctx.count_add(**dict(cols, end_pos=None))
    assert code == 0 and int(ctx.count_get()[0]) == int(want[0]) > 0
    # the oracle's own order checks (the host does them in the product)
    cols["pos"] = cols["pos"][::-1].copy()
    _, code, where = oracle.count_batch(**cols, n_chr=2, rchr=rchr, rstart=rstart, rend=rend)
    assert code == 1 and where > 0
    from seqkit_amd.capi import SeqkitHipError
    with pytest.raises(SeqkitHipError):
        ctx.count_set_regions(np.array([0, 2, 1], dtype=np.int32), rstart, rend)


# ---- f4: sam to fastq sequence() -----------------------------------------------------------------------------
def bam_rows(n, stride, seed, seq4_stride=None, iupac_frac=0.1):
    """Random BAM-style rows: packed 4-bit codes (mostly 1/2/4/8, some ambiguity codes), raw phred 0..60 with a few 255,
    ragged lengths including 0, flags with and without 0x10."""
    rng = np.random.default_rng(seed)
    seq4_stride = seq4_stride or (stride // 2 + 3) // 4 * 4
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, size=(n, seq4_stride * 2))]
    amb = rng.random((n, seq4_stride * 2)) < iupac_frac
    codes[amb] = rng.integers(0, 16, size=int(amb.sum()), dtype=np.uint8)
    seq4 = ((codes[:, 0::2] << 4) | codes[:, 1::2]).astype(np.uint8)
    qual = rng.integers(0, 61, size=(n, stride), dtype=np.uint8)
    qual[rng.random((n, stride)) < 0.01] = 255
    ln = rng.integers(0, stride + 1, size=n).astype(np.uint16)
    ln[rng.random(n) < 0.3] = stride
    flag = rng.choice(np.array([0, 16, 83, 99, 147, 163, 4, 1040], dtype=np.uint16), size=n)
    return np.ascontiguousarray(seq4), qual, ln, flag


def assert_rows_equal(got, want, ln):
    cols = np.arange(got.shape[1])[None, :]
    valid = cols < ln[:, None].astype(np.int64)
    assert np.array_equal(got[valid], want[valid])


def test_bam_sequence_kat_gpu(ctx, golden):
    from tests.test_oracle_kat import bam_sequence_case
    g = golden["bam_sequence"]
    for c in g["cases"]:
        seq4, qual, ln, flag = bam_sequence_case(c)
        out = ctx.bam_sequence(seq4, qual, ln, flag, g["min_baseq"])
        assert out[0, :ln[0]].tobytes() == c["out"].encode(), c


@pytest.fixture(params=["LDS tile where the pitch allows", "8-byte units everywhere"])
def seq_kernel(request, monkeypatch):
    """sequence() has two kernels (sk_kernels.hip): bam_sequence_tile_kernel for pitches that are a multiple of 4 up to 160,
    bam_sequence8_kernel for the rest — and for every pitch with SK_SEQ_TILE=0."""
    if request.param.startswith("8-byte"):
        monkeypatch.setenv("SK_SEQ_TILE", "0")
    return request.param


@pytest.mark.parametrize("n,stride,seq4_stride", [(1, 4, 4), (63, 8, 4), (65, 12, 8), (1000, 152, 76), (3000, 152, 80), (257, 100, 52), (130, 160, 80), (5000, 104, 52),
                                                  (70, 252, 128), (40, 40000, 20000), (2000, 148, 76), (129, 148, 80), (64, 156, 80), (100, 20, 12), (200, 12, 8)])
def test_bam_sequence_matches_oracle(ctx, oracle, seq_kernel, n, stride, seq4_stride):
    seq4, qual, ln, flag = bam_rows(n, stride, seed=n + stride, seq4_stride=seq4_stride)
    for m in (10, 0, 31, 200, 255):
        got = ctx.bam_sequence(seq4, qual, ln, flag, m)
        assert_rows_equal(got, oracle.bam_sequence_batch(seq4, qual, ln, flag, m), ln)
    got = ctx.bam_sequence(seq4, qual, None, flag, 10)               # len NULL: every row is full
    assert np.array_equal(got, oracle.bam_sequence_batch(seq4, qual, None, flag, 10))


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_bam_sequence(ctx, oracle, seq_kernel, seed):
    """Random row pitches (multiples of 8, and odd multiples of 4 whose last unit is clipped to its first dword), packed-row
    pitches with and without slack, row counts around the 64-row tile, any threshold, ragged lengths on both strands."""
    rng = np.random.default_rng(7000 + seed)
    stride = int(rng.choice([4, 8, 12, 16, 20, 24, 40, 56, 100, 104, 152, 248, 252, 256, 1000, 1024]))
    seq4_stride = (stride // 2 + 3) // 4 * 4 + int(rng.choice([0, 0, 4, 8]))
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 500, 3001]))
    seq4, qual, ln, flag = bam_rows(n, stride, seed=seed, seq4_stride=seq4_stride, iupac_frac=float(rng.choice([0.0, 0.1, 0.5])))
    m = int(rng.choice([0, 1, 10, 31, 60, 127, 128, 200, 255]))
    got = ctx.bam_sequence(seq4, qual, ln, flag, m)
    assert_rows_equal(got, oracle.bam_sequence_batch(seq4, qual, ln, flag, m), ln)
    if seed % 4 == 0:
        got = ctx.bam_sequence(seq4, qual, None, flag, m)
        assert np.array_equal(got, oracle.bam_sequence_batch(seq4, qual, None, flag, m))


def test_bam_sequence_every_length_and_strand(ctx, oracle, seq_kernel):
    """Every length 0..40 on both strands: the reverse path's funnel shifts and its partial last dword."""
    stride = 40
    lens = np.repeat(np.arange(0, 41, dtype=np.uint16), 2)
    seq4, qual, _, _ = bam_rows(len(lens), stride, seed=77, iupac_frac=0.2)
    flag = np.tile(np.array([0, 16], dtype=np.uint16), 41)
    got = ctx.bam_sequence(seq4, qual, lens, flag, 10)
    assert_rows_equal(got, oracle.bam_sequence_batch(seq4, qual, lens, flag, 10), lens)


def test_bam_sequence_multichunk_and_errors(ctx, oracle):
    seq4, qual, ln, flag = bam_rows(700000, 152, seed=5)           # > 128 MiB of staging: several chunks
    got = ctx.bam_sequence(seq4, qual, ln, flag, 10)
    assert_rows_equal(got, oracle.bam_sequence_batch(seq4, qual, ln, flag, 10), ln)
    from seqkit_amd.capi import SeqkitHipError
    with pytest.raises(SeqkitHipError):
        ctx.bam_sequence(np.zeros((2, 4), np.uint8), np.zeros((2, 10), np.uint8), None, np.zeros(2, np.uint16))     # stride not a multiple of 4
    with pytest.raises(SeqkitHipError):
        ctx.bam_sequence(np.zeros((2, 4), np.uint8), np.zeros((2, 12), np.uint8), None, np.zeros(2, np.uint16))     # seq4 rows too short


# ---- f3: barcode census ------------------------------------------------------------------------------------
def census_rows(strings, stride):
    m = np.zeros((len(strings), stride), dtype=np.uint8)
    for i, s in enumerate(strings):
        m[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    return m


def random_barcodes(n, L, stride, n_hot, hot_frac, seed, alphabet=b"ACGTN"):
    """n rows: a few dominant barcodes (the sheet), the rest random — the shape of a demultiplex dry run."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    hot = alpha[rng.integers(0, len(alpha), size=(n_hot, L))]
    m = np.zeros((n, stride), dtype=np.uint8)
    m[:, :L] = alpha[rng.integers(0, len(alpha), size=(n, L))]
    pick = rng.random(n) < hot_frac
    m[pick, :L] = hot[rng.integers(0, n_hot, size=int(pick.sum()))]
    return m


@pytest.fixture(params=["direct", "partition", "spilled_direct", "direct_tiny_front", "partition_tiny_front", "partition_front_inserts", "partition_long_way_only",
                        "direct_long_way_only"])
def census_path(request, monkeypatch):
    """The three ways a launch can take (sk_census.hip, census_add): keys the front tables have no room for are inserted by
    the front kernel itself (small launches), or written out and then partitioned + combined per table region, or — when
    most rows were written out — inserted as they lie.  Large launches choose between the last two by themselves; the
    environment forces each here so that the small cases below walk all of them.  In front of all three, rows whose bytes the
    workgroup's front table knows are counted there without building their key; `_tiny_front` shrinks that table to 64
    entries, so that even small inputs find both places of a string taken; `_front_inserts` makes the front kernel insert its
    table itself although the partition path is behind it (otherwise the table leaves as weighted records)."""
    if request.param.endswith("_tiny_front"):
        monkeypatch.setenv("SK_CENSUS_FRONT_ENTRIES", "64")
    if request.param.endswith("_front_inserts"):
        monkeypatch.setenv("SK_CENSUS_MERGE_RECORDS", "0")
    if request.param.endswith("_long_way_only"):                   # the fallback switch: no row takes the fast look at the front table
        monkeypatch.setenv("SK_CENSUS_LONG_WAY_ONLY", "1")
    if request.param.startswith("direct"):
        monkeypatch.setenv("SK_CENSUS_SPILL", "0")
    else:
        monkeypatch.setenv("SK_CENSUS_SPILL", "1")
        monkeypatch.setenv("SK_CENSUS_SPILL_MAX_PCT", "100" if request.param.startswith("partition") else "0")
    return request.param


def test_census_small_known(ctx, oracle, census_path):
    bc = census_rows([b"ACGT", b"ACG", b"ACGT", b"", b"TTTTTTTT", b"ACG", b"acgtn+AC", b"ACGT"], 8)
    ctx.census_reset()
    ctx.census_add(bc)
    got, total = ctx.census_entries()
    assert total == 5
    assert got == [(b"ACGT", 3, 0), (b"ACG", 2, 1), (b"", 1, 3), (b"TTTTTTTT", 1, 4), (b"acgtn+AC", 1, 6)]
    assert got == oracle.census(bc)
    st = ctx.census_stats()
    assert (st["distinct"], st["counted"], st["rejected"]) == (5, 8, 0)
    hist = ctx.census_count_hist()
    assert [int(x) for x in hist[:3]] == [3, 2, 0]            # counts 1,1,1 | 2,3
    assert ctx.census_entries(min_count=2)[0] == [(b"ACGT", 3, 0), (b"ACG", 2, 1)]
    got, total = ctx.census_entries(min_count=1, cap=2)       # cap < qualifying: total still says how many there are
    assert total == 5 and len(got) == 2
    ctx.census_reset()
    assert ctx.census_stats()["distinct"] == 0 and ctx.census_entries()[0] == []


@pytest.mark.parametrize("n,L,stride,n_hot,hot_frac", [(1, 8, 8, 1, 0.0), (255, 17, 17, 4, 0.5), (70000, 17, 24, 96, 0.9),
                                                       (300000, 8, 8, 96, 0.5), (200000, 31, 32, 10, 0.2), (100000, 6, 16, 3, 0.0)])
def test_census_matches_oracle(ctx, oracle, census_path, n, L, stride, n_hot, hot_frac):
    bc = random_barcodes(n, L, stride, n_hot, hot_frac, seed=n + L, alphabet=b"ACGTNacgtn+")
    ctx.census_reset()
    ctx.census_add(bc, L=L)
    got, total = ctx.census_entries()
    want = oracle.census(bc, L=L)
    assert total == len(want) and got == want
    st = ctx.census_stats()
    assert st["counted"] == n and st["rejected"] == 0 and sum(c for _, c, _ in got) == n


@pytest.mark.parametrize("stride", [8, 20, 21, 26, 27, 40, 41, 64])
def test_census_every_step_shape(ctx, oracle, census_path, stride):
    """A wave step is 4, 3, 2 or 1 tiles of 64 rows depending on the row pitch (5 KiB per step): row counts around the
    tile, step and workgroup boundaries of each, with and without assignment codes, ragged barcodes."""
    L = min(stride, 31)
    R = max(1, min(4, 5120 // (64 * stride)))
    rng = np.random.default_rng(stride)
    for n in (1, 63, 64, 65, 64 * R - 1, 64 * R, 64 * R + 1, 64 * R * 16 - 1, 64 * R * 16 + 1, 64 * R * 16 * 3 + 77):
        bc = random_barcodes(n, L, stride, 5, 0.6, seed=n + stride, alphabet=b"ACGTNacgtn+")
        short = rng.random(n) < 0.3                         # some barcodes end early: NUL padding inside L
        cut = rng.integers(0, L, size=n)
        for r in np.nonzero(short)[0]:
            bc[r, cut[r]:] = 0
        assign = np.where(rng.random(n) < 0.5, -1, rng.integers(0, 7, size=n)).astype(np.int32)
        for asg in (None, assign):
            ctx.census_reset()
            ctx.census_add(bc, L=L, assign=asg)
            got, total = ctx.census_entries()
            want = oracle.census(bc, L=L, assign=asg)
            assert total == len(want) and got == want, (stride, n, asg is not None)
            assert ctx.census_stats()["counted"] == (n if asg is None else int((asg == -1).sum()))


def test_census_only_unassigned_rows_batches_and_row_base(ctx, oracle, census_path):
    """The dry-run use (src/fasta_demultiplex.rs:190): only reads with no sample within max_diff are counted; fed in
    batches, in any order, the first-seen index is still the smallest global row."""
    n = 50000
    table = synth.make_sheet(96, 8, seed=5)
    bc, _ = synth.observe_barcodes(table, n, seed=6, p_exact=0.5, p_sub=0.2)
    ctx.set_barcodes(table, 1)
    assign = ctx.demux_assign(bc)[0]
    want = oracle.census(bc, assign=assign)
    assert 0 < len(want) < n
    ctx.census_reset()
    cuts = [0, 7, 20000, 20001, 43210, n]
    for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
        ctx.census_add(bc[lo:hi], assign=assign[lo:hi], row_base=lo)
    got, _ = ctx.census_entries()
    assert got == want
    assert ctx.census_stats()["counted"] == int((assign == -1).sum())


def test_census_rejects_foreign_bytes_and_bad_arguments(ctx, oracle, census_path):
    bc = census_rows([b"ACGT", b"AC-T", b"ACGU", b"ACGT", b"AC\0XX"], 5)       # bytes after the first NUL are padding
    ctx.census_reset()
    ctx.census_add(bc, L=5)
    st = ctx.census_stats()
    assert (st["distinct"], st["counted"], st["rejected"]) == (2, 3, 2)
    assert ctx.census_entries()[0] == [(b"ACGT", 2, 0), (b"AC", 1, 4)]
    from seqkit_amd.capi import SeqkitHipError
    with pytest.raises(SeqkitHipError):
        ctx.census_add(np.zeros((4, 40), dtype=np.uint8), L=32)


@pytest.mark.parametrize("shape", ["noisy", "one_hot_key", "mostly_new"])
def test_census_large_launch(ctx, oracle, monkeypatch, shape):
    """Large launches (8 M rows and more; 2 M here) take the partition path by themselves: a noisy dual-index run (keys repeat: partitioned
    and combined), a run where ONE barcode is a third of all rows yet cannot be in any front table (its bucket is split among
    workgroups) and a run of mostly new keys (more than half the rows are written out: inserted as they lie)."""
    monkeypatch.setenv("SK_CENSUS_SPILL_MIN_ROWS_LOG2", "21")
    n = 2_300_000
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    if shape == "noisy":
        bc, _ = synth.observe_barcodes(table, n, seed=12, halves=2)
    else:
        rng = np.random.default_rng(5)
        alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
        bc = np.zeros((n, 17), dtype=np.uint8)
        bc[:, :17] = alpha[rng.integers(0, 4, size=(n, 17))]                   # new keys throughout ...
        if shape == "one_hot_key":
            few = alpha[rng.integers(0, 4, size=(5000, 17))]                   # ... or 5 000 keys: more than a front table holds,
            bc[:, :17] = few[rng.integers(0, 5000, size=n)]
            bc[rng.random(n) < 0.33] = np.frombuffer(b"GATTACAG+CATCATCA", dtype=np.uint8)      # and one that is everywhere — but first seen
            bc[:300000, :17] = few[rng.integers(0, 5000, size=300000)]         # only after the front tables have filled up
    ctx.census_reset()
    ctx.census_add(bc, L=17)
    got, total = ctx.census_entries(min_count=2 if shape == "mostly_new" else 1)
    want = oracle.census(bc, L=17)
    if shape == "mostly_new":
        want = [w for w in want if w[1] >= 2]
    assert total == len(want) and got == want
    st = ctx.census_stats()
    assert st["counted"] == n and st["rejected"] == 0


@pytest.mark.parametrize("L,stride", [(4, 4), (7, 9), (17, 17), (24, 24), (31, 32)])
def test_census_every_byte_value_at_every_position(ctx, oracle, census_path, L, stride):
    """The key is built four characters at a time with packed-byte arithmetic (sk_census.hip, census_classify): every byte
    value at every position of the barcode (so at every byte of a dword, at every alignment of the row in the tile) — counted
    iff it is one of "ACGTNacgtn+", a NUL ends the barcode, anything else rejects the row."""
    rng = np.random.default_rng(L)
    alpha = np.frombuffer(b"ACGTNacgtn+", dtype=np.uint8)
    rows = []
    for pos in range(L):
        base = alpha[rng.integers(0, len(alpha), size=(256, L))]
        base[:, pos] = np.arange(256, dtype=np.uint8)
        rows.append(base)
    bc = np.zeros((256 * L, stride), dtype=np.uint8)
    bc[:, :L] = np.concatenate(rows)
    bc[:, L:] = 0x7F                                        # what lies beyond L is never looked at
    ok = np.isin(np.arange(256, dtype=np.uint8), np.concatenate([alpha, np.zeros(1, np.uint8)]))
    valid = np.tile(ok, L)                                  # the rows the device counts (the hosts count the others themselves)
    ctx.census_reset()
    ctx.census_add(bc, L=L)
    got, total = ctx.census_entries()
    want = oracle.census(bc, L=L, assign=np.where(valid, -1, 0).astype(np.int32))
    st = ctx.census_stats()
    assert total == len(want) and got == want
    assert st["rejected"] == L * (256 - len(alpha) - 1) and st["counted"] == L * (len(alpha) + 1)


def test_census_anything_after_the_first_nul_is_padding(ctx, oracle, census_path):
    """A NUL at every position, followed by bytes of every kind (other NULs, bytes outside the alphabet, letters): the barcode
    is what precedes the first NUL, whatever follows (src/fasta_demultiplex.rs:190 counts the header's barcode string; the
    matrix rows are NUL-padded by the packer, but the C-ABI does not ask for clean padding)."""
    L, stride, n = 23, 27, 6000
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ACGTNacgtn+", dtype=np.uint8)
    bc = np.zeros((n, stride), dtype=np.uint8)
    bc[:, :L] = alpha[rng.integers(0, len(alpha), size=(n, L))]
    cut = rng.integers(0, L, size=n)
    for r in range(n):
        bc[r, cut[r]] = 0
        bc[r, cut[r] + 1:] = rng.integers(0, 256, size=stride - cut[r] - 1)
    bc[::7, :L] = alpha[rng.integers(0, len(alpha), size=(len(bc[::7]), L))]      # and rows without a NUL among them
    ctx.census_reset()
    ctx.census_add(bc, L=L)
    got, total = ctx.census_entries()
    want = oracle.census(bc, L=L)
    assert total == len(want) and got == want
    assert ctx.census_stats()["rejected"] == 0 and ctx.census_stats()["counted"] == n


@pytest.mark.parametrize("seed", range(int(os.environ.get("SK_FUZZ_SEEDS", "24"))))      # SK_FUZZ_SEEDS=400: the long run, once per round on the GPU box
def test_fuzz_census(ctx, oracle, monkeypatch, seed):
    """Random shapes through a random path: length, pitch, row count, alphabet, share and number of frequent barcodes, early
    NULs, bytes outside the alphabet, assignment codes, batches with row_base — every distinct barcode, its count and its
    first row against the oracle's map."""
    rng = np.random.default_rng(1000 + seed)
    path = ("direct", "partition", "spilled_direct")[seed % 3]
    monkeypatch.setenv("SK_CENSUS_SPILL", "0" if path == "direct" else "1")
    monkeypatch.setenv("SK_CENSUS_SPILL_MAX_PCT", "0" if path == "spilled_direct" else "100")
    if seed % 4 == 3:
        monkeypatch.setenv("SK_CENSUS_FRONT_ENTRIES", "64")  # a front table that is full at once: both places of most strings taken
    if seed % 5 == 4:
        monkeypatch.setenv("SK_CENSUS_MERGE_RECORDS", "0")   # the front kernel inserts its table itself
    if seed % 7 == 6:
        monkeypatch.setenv("SK_CENSUS_LONG_WAY_ONLY", "1")   # the fallback switch
    L = int(rng.integers(1, 32))
    stride = int(rng.integers(L, min(64, L + 9) + 1))
    n = int(rng.integers(1, 250_000))
    alpha = np.frombuffer((b"ACGT", b"ACGTN", b"ACGTNacgtn+")[int(rng.integers(0, 3))], dtype=np.uint8)
    n_hot = int(rng.integers(1, 3000))
    bc = random_barcodes(n, L, stride, n_hot, float(rng.random()), seed=seed, alphabet=alpha.tobytes())
    bc[:, L:] = rng.integers(0, 256, size=(n, stride - L), dtype=np.uint8)      # never looked at
    short = rng.random(n) < rng.random() * 0.5
    cut = rng.integers(0, L, size=n)
    for r in np.nonzero(short)[0]:
        bc[r, cut[r]:L] = 0
    foreign = rng.random(n) < 0.01
    pos = rng.integers(0, L, size=n)
    bc[foreign, pos[foreign]] = rng.choice(np.frombuffer(b"-_.UXxz0 \x7f\xff\x01", dtype=np.uint8), size=int(foreign.sum()))
    # the rows the device counts: no byte outside the alphabet before the first NUL (within L)
    inside = np.isin(bc[:, :L], np.concatenate([alpha if len(alpha) == 11 else np.frombuffer(b"ACGTNacgtn+", np.uint8), np.zeros(1, np.uint8)]))
    nul = bc[:, :L] == 0
    before_nul = np.cumsum(nul, axis=1) == 0
    valid = ~((~inside) & before_nul).any(axis=1)
    assign = None
    if rng.random() < 0.5:
        assign = np.where(rng.random(n) < 0.6, -1, rng.integers(0, 5, size=n)).astype(np.int32)
    counted = valid if assign is None else valid & (assign == -1)
    base = int(rng.integers(0, 1 << 40))
    ctx.census_reset()
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 4)))]))
    for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:]))):
        ctx.census_add(bc[lo:hi], L=L, assign=None if assign is None else assign[lo:hi], row_base=base + lo)
    got, total = ctx.census_entries()
    want = oracle.census(bc, L=L, assign=np.where(counted, -1, 0).astype(np.int32))
    assert total == len(want) and got == [(b, c, f + base) for b, c, f in want], (seed, path, L, stride, n)
    st = ctx.census_stats()
    assert st["counted"] == int(counted.sum())
    assert st["rejected"] == int(((~valid) if assign is None else (~valid) & (assign == -1)).sum())


def test_census_grows_past_its_first_table(oracle, monkeypatch):
    """More distinct barcodes than half the table: it is rehashed between launches, nothing is lost.  (The first table
    normally has 2^26 slots; a context created with SK_CENSUS_SLOTS_LOG2=16 starts small.)"""
    import seqkit_amd
    monkeypatch.setenv("SK_CENSUS_SLOTS_LOG2", "16")
    n = 3_000_000
    rng = np.random.default_rng(11)
    vals = rng.integers(0, 4 ** 11, size=n, dtype=np.int64)                   # ~2.1 M distinct 11-mers
    digits = (vals[:, None] >> (2 * np.arange(11))) & 3
    bc = np.zeros((n, 16), dtype=np.uint8)
    bc[:, :11] = np.frombuffer(b"ACGT", dtype=np.uint8)[digits]
    with seqkit_amd.Context(0) as c2:
        c2.census_reset()
        slots0 = c2.census_stats()["slots"]
        assert slots0 == 1 << 16
        for lo in range(0, n, 700_000):                                       # several host batches, each several launches
            c2.census_add(bc[lo:lo + 700_000], L=11, row_base=lo)
        st = c2.census_stats()
        uniq, first, counts = np.unique(vals, return_index=True, return_counts=True)
        assert st["distinct"] == len(uniq) and st["counted"] == n and st["slots"] > slots0 and st["slots"] >= 2 * st["distinct"]
        got, total = c2.census_entries(min_count=3)
        order = np.argsort(first[counts >= 3], kind="stable")
        assert total == int((counts >= 3).sum()) > 1000
        assert [(c, f) for _, c, f in got] == [(int(c), int(f)) for c, f in zip(counts[counts >= 3][order], first[counts >= 3][order])]
        hist = c2.census_count_hist()
        assert int(hist.sum()) == len(uniq) and int(hist[0]) == int((counts == 1).sum())


@pytest.mark.parametrize("chunk_log2,min_log2", [("15", "12"), ("17", "17"), ("13", "6")])
def test_census_grows_between_the_launches_of_one_call(oracle, monkeypatch, chunk_log2, min_log2):
    """One sk_census_add call that is many launches: with all-distinct barcodes and a small first table every launch has to
    decide between a smaller bite and a bigger table — not only the last one.  The table never gets more than half full
    (no row lost to a probe overflow, which census_stats would report as an error), whatever the launch sizes are."""
    import seqkit_amd
    monkeypatch.setenv("SK_CENSUS_SLOTS_LOG2", "14")
    monkeypatch.setenv("SK_CENSUS_CHUNK_LOG2", chunk_log2)
    monkeypatch.setenv("SK_CENSUS_MIN_CHUNK_LOG2", min_log2)
    n = 600_000
    vals = np.random.default_rng(21).permutation(4 ** 10)[:n].astype(np.int64)      # all distinct 10-mers
    digits = (vals[:, None] >> (2 * np.arange(10))) & 3
    bc = np.zeros((n, 12), dtype=np.uint8)
    bc[:, :10] = np.frombuffer(b"ACGT", dtype=np.uint8)[digits]
    with seqkit_amd.Context(0) as c2:
        c2.census_reset()
        c2.census_add(bc, L=10)
        st = c2.census_stats()                                # raises if any row overflowed its probe budget
        assert st["distinct"] == n and st["counted"] == n and st["slots"] >= 2 * n
        c2.census_add(bc[:1000], L=10, row_base=n)            # all of them again: counts 2, first rows unchanged
        got, total = c2.census_entries(min_count=2)
        assert total == 1000 and [(c, f) for _, c, f in got] == [(2, i) for i in range(1000)]


@pytest.mark.parametrize("chunk_log2", ["12", "15"])
def test_host_entry_points_in_many_chunks(ctx, oracle, monkeypatch, chunk_log2):
    """Every host-pointer entry point cut into many chunks (SK_HOST_CHUNK_LOG2): the two-lane pipeline — alternating streams,
    alternating workspace halves, accumulators shared by both lanes — gives what one chunk gives."""
    monkeypatch.setenv("SK_HOST_CHUNK_LOG2", chunk_log2)
    n, L = 5003, 101
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=31, halves=2)
    ln = synth.ragged_lengths(n, L, seed=31)
    mates = []
    for mi in range(2):
        seq, qual = synth.make_reads(n, L, seed=230 + mi)
        mates.append((seq, synth.add_forced_classes(qual, seed=240 + mi), ln))
    ctx.set_barcodes(table, 1)
    r = ctx.fused_pass(mates, 20, bc=bc, want_detail=True)
    e = oracle.demux_batch(table, bc, 1)
    assert np.array_equal(r["assign"], e[0]) and np.array_equal(r["lowest_diff"], e[1]) and np.array_equal(ctx.counts(), e[4])
    for mi, (seq, qual, _) in enumerate(mates):
        assert np.array_equal(r["lowest_k"][mi], oracle.trim_batch(qual, ln, 20))
        exp = oracle.mask_batch(seq, qual, ln, 20)
        valid = np.arange(L)[None, :] < ln[:, None]
        assert np.array_equal(r["out_seq"][mi][valid], exp[valid])
    assert np.array_equal(ctx.mask_by_quality(mates[0][0], mates[0][1], None, 30), oracle.mask_batch(mates[0][0], mates[0][1], None, 30))
    assert np.array_equal(ctx.trim_by_quality(mates[1][1], ln, 2), oracle.trim_batch(mates[1][1], ln, 2))
    assert np.array_equal(ctx.demux_assign(bc, want_detail=False)[0], e[0])
    # BAM columns
    flag, tid, mtid, tlen = synth.make_bam_cores(70001, seed=9)
    counters, hist, total = ctx.bam_flag_tlen(flag, tid, mtid, tlen, 5000)
    ec, eh, et = oracle.bam_flag_tlen(flag, tid, mtid, tlen, 5000)
    assert np.array_equal(counters, ec) and np.array_equal(hist, eh) and total == et
    keep, kept = ctx.bam_fragments(flag, tid, mtid, tlen, 50, 700)
    ek = oracle.fragments_keep(flag, tid, mtid, tlen, 50, 700)
    assert np.array_equal(keep, ek) and kept == int(ek.sum())
    seq4, qual, lens, flags = bam_rows(3001, 152, seed=6)
    assert_rows_equal(ctx.bam_sequence(seq4, qual, lens, flags, 10), oracle.bam_sequence_batch(seq4, qual, lens, flags, 10), lens)
    # census
    ctx.census_reset()
    ctx.census_add(bc)
    got, _ = ctx.census_entries()
    assert got == oracle.census(bc)
    # sam count
    n_chr = 5
    cols, rchr, rstart, rend = count_inputs(60_000, n_chr, 3000, seed=12)
    want, code, _ = oracle.count_batch(**cols, n_chr=n_chr, rchr=rchr, rstart=rstart, rend=rend)
    chr_off, gs, ge, gi = grouped_regions(rchr, rstart, rend, n_chr)
    ctx.count_set_regions(chr_off, gs, ge, gi, n_regions=3000)
    ctx.count_add(**dict(cols, end_pos=None))
    assert code == 0 and np.array_equal(ctx.count_get()[gi], want[gi])


def test_errors_are_codes_not_crashes(ctx):
    import seqkit_amd
    with pytest.raises(seqkit_amd.SeqkitHipError):
        ctx.set_barcodes(np.zeros((2, 300), dtype=np.uint8))          # L > 255
    ctx.set_barcodes(np.array([list(b"ACGTACGT")], dtype=np.uint8))
    with pytest.raises(seqkit_amd.SeqkitHipError):
        ctx.demux_assign(np.zeros((4, 5), dtype=np.uint8))            # bc_stride < L
    with pytest.raises(seqkit_amd.SeqkitHipError):
        seqkit_amd.Context(4096)                                      # no such device


@pytest.mark.gpu
@pytest.mark.parametrize("gather_form", [False, True])
@pytest.mark.parametrize("shape", ["16 x 8", "96 x 8+8", "384 x 8+8", "130 x 12"])
def test_many_batches_in_one_call_equal_the_single_calls_and_the_oracle(ctx, oracle, shape, gather_form, monkeypatch):
    """sk_demux_assign_many_dev: ragged batches (empty ones among them), with and without the detail columns of
    matched rows — every batch's outputs are what its own call gives and what the oracle says (src/fasta_demultiplex.rs:154-194), the
    counters are the sum over the batches.  gather_form: the gathered-row kernel's one-launch form too (SK_LUT_MANY_GATHER=1; the
    product runs those sheets' batches back to back, which is faster)."""
    from seqkit_amd import capi, synth
    if gather_form:
        monkeypatch.setenv("SK_LUT_MANY_GATHER", "1")
    S = int(shape.split(" x ")[0])
    dual = "+" in shape
    L = int(shape.split(" x ")[1].split("+")[0])
    table = synth.make_sheet(S, L, dual=dual, seed=S + 2)
    sizes = [70_001, 0, 1, 255, 256, 300_000, 64, 12_345, 1_500_001]
    bcs = [synth.observe_barcodes(table, max(n, 1), seed=100 + i, halves=2 if dual else 1)[0][:n] for i, n in enumerate(sizes)]
    ctx.set_barcodes(table, 1)
    for detail in (False, True):
        ctx.set_detail_mode(capi.SK_DETAIL_MATCHED if detail else capi.SK_DETAIL_FULL)
        ctx.counts_reset()
        ptrs, batches = [], []
        try:
            for bc in bcs:
                n = len(bc)
                d_bc = ctx.malloc_device(bc.nbytes + 64)
                outs = [ctx.malloc_device(max(n, 1) * w + 64) for w in ((4, 1, 2, 2) if detail else (4,))]
                ptrs += [d_bc] + outs
                if n:
                    ctx.copy_h2d(d_bc, np.ascontiguousarray(bc))
                batches.append((d_bc, n, *outs))
            ctx.sync()
            ctx.demux_assign_many_dev(batches, bcs[0].shape[1])
            ctx.sync()
            total = np.zeros(S + 3, dtype=np.uint64)
            for bc, b in zip(bcs, batches):
                n = len(bc)
                if n == 0:
                    continue
                e = oracle.demux_batch(table, bc, 1)
                total += e[4].astype(np.uint64)
                got = np.empty(n, dtype=np.int32)
                ctx.copy_d2h(got, b[2])
                ctx.sync()
                assert np.array_equal(got, e[0])
                if detail:
                    m = e[0] != -1
                    for k, (t, col) in enumerate(((np.uint8, 1), (np.int16, 2), (np.int16, 3))):
                        g = np.empty(n, dtype=t)
                        ctx.copy_d2h(g, b[3 + k])
                        ctx.sync()
                        assert np.array_equal(g[m], e[col][m])
            assert np.array_equal(ctx.counts(), total)
        finally:
            ctx.set_detail_mode(capi.SK_DETAIL_FULL)
            for p in ptrs:
                ctx.free_device(p)


@pytest.mark.gpu
@pytest.mark.parametrize("two_streams", ["1", "0"])
def test_trim_many_batches_equal_the_oracle(ctx, oracle, two_streams, monkeypatch):
    """sk_trim_by_quality_many_dev: batches of cfg 2's classes with and without row lengths (src/fasta_trim_by_quality.rs:28-42), their
    launches on the ctx's two streams in turn (the default) and on one."""
    from seqkit_amd import synth
    monkeypatch.setenv("SK_MANY_TWO_STREAMS", two_streams)
    stride = 150
    ptrs, batches, expect = [], [], []
    try:
        for i, n in enumerate([40_000, 1, 0, 64, 100_001, 777]):
            _, qual = synth.make_reads(max(n, 1), stride, seed=30 + i)
            qual = synth.add_forced_classes(qual, seed=40 + i)[:n]
            ln = None
            if i % 2:
                ln = np.random.default_rng(i).integers(0, stride + 1, size=n).astype(np.uint16)
            d_q, d_k = ctx.malloc_device(qual.nbytes + 64), ctx.malloc_device(2 * max(n, 1) + 64)
            d_l = ctx.malloc_device(2 * max(n, 1) + 64) if ln is not None else 0
            ptrs += [p for p in (d_q, d_k, d_l) if p]
            if n:
                ctx.copy_h2d(d_q, np.ascontiguousarray(qual))
                if ln is not None:
                    ctx.copy_h2d(d_l, ln)
            batches.append((d_q, d_l, n, d_k))
            expect.append(oracle.trim_batch(qual, ln, 20) if n else None)
        ctx.sync()
        ctx.trim_by_quality_many_dev(batches, stride, 20)
        ctx.sync()
        for (d_q, d_l, n, d_k), e in zip(batches, expect):
            if n == 0:
                continue
            got = np.empty(n, dtype=np.uint16)
            ctx.copy_d2h(got, d_k)
            ctx.sync()
            assert np.array_equal(got, e)
    finally:
        for p in ptrs:
            ctx.free_device(p)
