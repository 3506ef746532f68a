"""ctypes view of oracle/liboracle.so — the plain-C restatement of the reference's loops.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from seqkit_amd/.  PARITY UNPINNED: the reference is Rust, cannot be
built here and has no tests; the C is checked against hand-derived
known-answer vectors only (tests/golden/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
FASTA_BIN = os.path.join(HERE, "fasta_oracle")
SAM_BIN = os.path.join(HERE, "sam_oracle")
NONE, AMBIGUOUS = -1, -2
_lib = None


def build(force: bool = False) -> None:
    srcs = ["seqkit_oracle.c", "seqkit_oracle.h", "cli_common.h", "fasta_oracle_main.c", "sam_oracle_main.c", "Makefile"]
    outs = [LIB, FASTA_BIN, SAM_BIN]
    newest = max(os.path.getmtime(os.path.join(HERE, s)) for s in srcs)
    if force or not all(os.path.exists(o) and os.path.getmtime(o) >= newest for o in outs):
        r = subprocess.run(["make", "-C", HERE, "-B", "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        vp, i32, i64, u8 = C.c_void_p, C.c_int, C.c_int64, C.c_uint8
        _lib.orc_trim_lowest_k.restype = C.c_uint32
        _lib.orc_trim_lowest_k.argtypes = [vp, C.c_uint32, u8]
        _lib.orc_mask_bytes.restype = None
        _lib.orc_mask_bytes.argtypes = [vp, vp, C.c_uint32, u8]
        _lib.orc_barcode_diff.restype = C.c_size_t
        _lib.orc_barcode_diff.argtypes = [vp, vp, C.c_size_t]
        _lib.orc_trim_batch.restype = None
        _lib.orc_trim_batch.argtypes = [vp, vp, i32, i64, u8, vp]
        _lib.orc_mask_batch.restype = None
        _lib.orc_mask_batch.argtypes = [vp, vp, vp, i32, i64, u8]
        _lib.orc_demux_batch.restype = None
        _lib.orc_demux_batch.argtypes = [vp, i32, i32, i32, vp, i32, i64, vp, vp, vp, vp, vp]
        _lib.orc_bam_flag_tlen.restype = None
        _lib.orc_bam_flag_tlen.argtypes = [vp, vp, vp, vp, i64, C.c_int32, vp, vp, vp]
        _lib.orc_fragment_lengths_stop.restype = i64
        _lib.orc_fragment_lengths_stop.argtypes = [vp, vp, vp, vp, i64, C.c_int32, C.c_uint64, vp, vp]
        _lib.orc_fragments_keep.restype = i64
        _lib.orc_fragments_keep.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp]
        _lib.orc_gc_count.restype = None
        _lib.orc_gc_count.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib.orc_count_state_init.restype = None
        _lib.orc_count_state_init.argtypes = [vp]
        _lib.orc_count_state_free.restype = None
        _lib.orc_count_state_free.argtypes = [vp]
        _lib.orc_count_batch.restype = i32
        _lib.orc_count_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, C.c_int32, vp, vp, vp, i64, vp, vp]
        _lib.orc_bam_sequence_batch.restype = None
        _lib.orc_bam_sequence_batch.argtypes = [vp, i32, vp, i32, vp, vp, i64, u8, vp]
        _lib.orc_census.restype = i64
        _lib.orc_census.argtypes = [vp, i32, i32, i64, vp, i64, vp]
        _lib.orc_free.restype = None
        _lib.orc_free.argtypes = [vp]
        _lib.orc_trim_end_len.restype = C.c_size_t
        _lib.orc_trim_end_len.argtypes = [C.c_char_p, C.c_size_t]
        _lib.orc_trim_start_off.restype = C.c_size_t
        _lib.orc_trim_start_off.argtypes = [C.c_char_p, C.c_size_t]
        _lib.orc_utf8_valid.restype = i32
        _lib.orc_utf8_valid.argtypes = [C.c_char_p, C.c_size_t]
        _lib.orc_find_bc_field_stats.restype = i32
        _lib.orc_find_bc_field_stats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _lib.orc_find_bc_field.restype = i32
        _lib.orc_find_bc_field.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


def trim_lowest_k(qual: bytes, min_baseq: int) -> int:
    buf = np.frombuffer(qual, dtype=np.uint8) if len(qual) else np.zeros(1, dtype=np.uint8)
    return int(lib().orc_trim_lowest_k(_p(buf), len(qual), min_baseq))


def mask_bytes(seq: bytes, qual: bytes, min_baseq: int) -> bytes:
    s = np.frombuffer(seq, dtype=np.uint8).copy()
    q = np.frombuffer(qual, dtype=np.uint8)
    lib().orc_mask_bytes(_p(s), _p(q), len(seq), min_baseq)
    return s.tobytes()


def barcode_diff(obs: bytes, cand: bytes) -> int:
    assert len(obs) == len(cand)
    o = np.frombuffer(obs, dtype=np.uint8)
    c = np.frombuffer(cand, dtype=np.uint8)
    return int(lib().orc_barcode_diff(_p(o), _p(c), len(obs)))


def trim_batch(qual: np.ndarray, length, min_baseq: int) -> np.ndarray:
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    n, stride = qual.shape
    out = np.empty(n, dtype=np.uint16)
    lib().orc_trim_batch(_p(qual), _p(length), stride, n, min_baseq, _p(out))
    return out


def mask_batch(seq: np.ndarray, qual: np.ndarray, length, min_baseq: int) -> np.ndarray:
    seq = np.ascontiguousarray(seq, dtype=np.uint8).copy()
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    n, stride = seq.shape
    lib().orc_mask_batch(_p(seq), _p(qual), _p(length), stride, n, min_baseq)
    return seq


def demux_batch(table: np.ndarray, bc: np.ndarray, max_diff: int = 1):
    table = np.ascontiguousarray(table, dtype=np.uint8)
    bc = np.ascontiguousarray(bc, dtype=np.uint8)
    S, L = table.shape
    n, bstride = bc.shape
    assign = np.empty(n, dtype=np.int32)
    low = np.empty(n, dtype=np.uint8)
    first = np.empty(n, dtype=np.int16)
    last = np.empty(n, dtype=np.int16)
    counts = np.zeros(S + 3, dtype=np.uint64)
    lib().orc_demux_batch(_p(table) if S else None, S, L, max_diff, _p(bc), bstride, n, _p(assign), _p(low), _p(first),
                          _p(last), _p(counts))
    return assign, low, first, last, counts


def bam_flag_tlen(flag, tid, mtid, tlen, max_frag: int = 5000):
    n = len(flag)
    counters = np.zeros(3, dtype=np.uint64)
    hist = np.zeros(max_frag + 1, dtype=np.uint64)
    total = np.zeros(1, dtype=np.uint64)
    lib().orc_bam_flag_tlen(_p(flag), _p(tid), _p(mtid), _p(tlen), n, max_frag, _p(counters), _p(hist), _p(total))
    return counters, hist, int(total[0])


def fragments_keep(flag, tid, mtid, tlen, min_size: int = 0, max_size: int = 5000) -> np.ndarray:
    n = len(flag)
    keep = np.zeros(n, dtype=np.uint8)
    lib().orc_fragments_keep(_p(flag), _p(tid), _p(mtid), _p(tlen), n, min_size, max_size, _p(keep))
    return keep


def trim_end_len(s: bytes) -> int:
    return int(lib().orc_trim_end_len(s, len(s)))


def utf8_valid(s: bytes) -> bool:
    return bool(lib().orc_utf8_valid(s, len(s)))


def find_bc_field(h: bytes):
    st, en = C.c_size_t(), C.c_size_t()
    if lib().orc_find_bc_field(h, len(h), C.byref(st), C.byref(en)):
        return int(st.value), int(en.value)
    return None


def find_bc_field_stats(h: bytes):
    st, en = C.c_size_t(), C.c_size_t()
    if lib().orc_find_bc_field_stats(h, len(h), C.byref(st), C.byref(en)):
        return int(st.value), int(en.value)
    return None


def census(bc: np.ndarray, L: int | None = None, assign=None, row_base: int = 0):
    """[(barcode bytes, count, first_row)] in first-seen order (f3)."""
    bc = np.ascontiguousarray(bc, dtype=np.uint8)
    n, stride = bc.shape
    if assign is not None:
        assign = np.ascontiguousarray(assign, dtype=np.int32)
    out = C.c_void_p()
    k = lib().orc_census(_p(bc), stride, stride if L is None else L, n, _p(assign) if assign is not None else None, row_base,
                         C.byref(out))
    if k < 0:
        raise RuntimeError("orc_census failed")
    dt = np.dtype([("barcode", "S32"), ("count", np.uint64), ("first_row", np.int64)])
    res = []
    if k:
        arr = np.frombuffer((C.c_char * (k * dt.itemsize)).from_address(out.value), dtype=dt).copy()
        res = [(bytes(e["barcode"]), int(e["count"]), int(e["first_row"])) for e in arr]
    lib().orc_free(out)
    return res


def bam_sequence_batch(seq4: np.ndarray, qual: np.ndarray, length, flag: np.ndarray, min_baseq: int = 10) -> np.ndarray:
    """f4: ASCII bases [n, stride] (bytes past a row's length are left 0)."""
    seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    flag = np.ascontiguousarray(flag, dtype=np.uint16)
    n, stride = qual.shape
    ln = None if length is None else np.ascontiguousarray(length, dtype=np.uint16)
    out = np.zeros((n, stride), dtype=np.uint8)
    lib().orc_bam_sequence_batch(_p(seq4), seq4.shape[1], _p(qual), stride, _p(ln) if ln is not None else None, _p(flag), n, min_baseq, _p(out))
    return out


class _CountParams(C.Structure):
    _fields_ = [("min_mapq", C.c_uint8), ("max_frag_len", C.c_uint32), ("single_end", C.c_int), ("count_centers", C.c_int)]


class _CountState(C.Structure):
    _fields_ = [("prev_chr", C.c_int32), ("prev_pos", C.c_int64), ("deque", C.c_void_p), ("front", C.c_int64), ("len", C.c_int64), ("cap", C.c_int64)]


def count_batch(flag, mapq, tid, mtid, pos, mpos, tlen, end_pos, n_chr, rchr, rstart, rend, min_mapq=0, max_frag_len=5000, single_end=False,
                center=False):
    """src/sam_count.rs:44-127 over the records in order: (region_frags, code, where); code 1 = not coordinate sorted."""
    n = len(flag)
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    flag, mapq = a(flag, np.uint16), a(mapq, np.uint8)
    tid, mtid, pos, mpos, tlen = (a(x, np.int32) for x in (tid, mtid, pos, mpos, tlen))
    end_pos = None if end_pos is None else a(end_pos, np.int32)
    rchr, rstart, rend = a(rchr, np.int32), a(rstart, np.uint32), a(rend, np.uint32)
    frags = np.zeros(max(len(rstart), 1), dtype=np.uint32)
    p = _CountParams(min_mapq, max_frag_len, int(single_end), int(center))
    st = _CountState()
    lib().orc_count_state_init(C.byref(st))
    where = C.c_int64(-1)
    code = lib().orc_count_batch(C.byref(st), _p(flag), _p(mapq), _p(tid), _p(mtid), _p(pos), _p(mpos), _p(tlen),
                                 _p(end_pos) if end_pos is not None else None, n, C.byref(p), n_chr, _p(rchr), _p(rstart), _p(rend),
                                 len(rstart), _p(frags), C.byref(where))
    lib().orc_count_state_free(C.byref(st))
    return frags[:len(rstart)], code, int(where.value)


def gc_count(seq: bytes):
    """(gc, total) of src/fasta_gc_content.rs:44-45."""
    g, t = C.c_uint64(), C.c_uint64()
    buf = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, dtype=np.uint8)
    lib().orc_gc_count(_p(buf), len(seq), C.byref(g), C.byref(t))
    return int(g.value), int(t.value)
