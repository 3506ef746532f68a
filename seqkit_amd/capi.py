"""ctypes view of include/seqkit_hip.h.

Nothing here computes: every method marshals numpy arrays (host entry points) or raw device
addresses (``*_dev`` entry points, e.g. ``torch.Tensor.data_ptr()``) into the C-ABI.  There is
deliberately no fallback path: if libseqkit_hip.so is absent or no gfx950 GPU is visible the
constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build

SK_ASSIGN_NONE = -1
SK_ASSIGN_AMBIGUOUS = -2

# every symbol include/seqkit_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "sk_version", "sk_device_count", "sk_create", "sk_destroy", "sk_last_error", "sk_sync", "sk_stream",
    "sk_malloc_device", "sk_free_device", "sk_malloc_pinned", "sk_free_pinned", "sk_copy_h2d", "sk_copy_d2h",
    "sk_set_barcodes", "sk_set_detail_mode", "sk_barcode_table_info", "sk_demux_assign", "sk_demux_assign_dev", "sk_trim_by_quality", "sk_trim_by_quality_dev",
    "sk_mask_by_quality", "sk_mask_by_quality_dev", "sk_fused_pass", "sk_fused_pass_dev",
    "sk_fused_pass_many_dev", "sk_demux_assign_many_dev", "sk_trim_by_quality_many_dev",
    "sk_blocked_layout_init", "sk_fused_pass_blocked_dev", "sk_fused_tune_placement_dev",
    "sk_counts_reset", "sk_counts_get", "sk_counts_device_ptr",
    "sk_comm_ready", "sk_comm_get_unique_id", "sk_comm_init_rank", "sk_comm_destroy", "sk_counts_allreduce", "sk_allreduce_u64_dev", "sk_bam_flag_tlen", "sk_bam_flag_tlen_dev",
    "sk_bgzf_deflate", "sk_bgzf_deflate_dev", "sk_bgzf_inflate_dev", "sk_bam_walk_dev", "sk_bam_walk_reduce_dev", "sk_bam_file_reduce",
    "sk_bam_fragments", "sk_bam_fragments_dev", "sk_bam_sequence", "sk_bam_sequence_dev",
    "sk_count_set_regions", "sk_count_add", "sk_count_add_dev", "sk_count_get", "sk_gc_set_genome", "sk_gc_count",
    "sk_census_reset", "sk_census_add", "sk_census_add_dev", "sk_census_stats", "sk_census_count_hist", "sk_census_entries",
    "sk_timer_start", "sk_timer_stop",
]


class SeqkitHipError(RuntimeError):
    pass


class _Mate(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p),
                ("out_seq", C.c_void_p), ("lowest_k", C.c_void_p)]


class _FusedArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_mates", C.c_int), ("stride", C.c_int), ("min_baseq", C.c_uint8),
                ("mate", _Mate * 2), ("bc", C.c_void_p), ("bc_stride", C.c_int), ("assign", C.c_void_p),
                ("lowest_diff", C.c_void_p), ("first_idx", C.c_void_p), ("last_idx", C.c_void_p),
                ("counts", C.c_void_p)]


class _DemuxBatch(C.Structure):
    _fields_ = [("bc", C.c_void_p), ("n", C.c_int64), ("assign", C.c_void_p), ("lowest_diff", C.c_void_p), ("first_idx", C.c_void_p), ("last_idx", C.c_void_p)]


class _TrimBatch(C.Structure):
    _fields_ = [("qual", C.c_void_p), ("len", C.c_void_p), ("n", C.c_int64), ("lowest_k", C.c_void_p)]


class _FusedCandidates(C.Structure):
    _fields_ = [("k", C.c_int), ("seq", (C.c_void_p * 8) * 2), ("qual", (C.c_void_p * 8) * 2), ("out_seq", (C.c_void_p * 8) * 2)]


SK_BLK_MASK, SK_BLK_TRIM, SK_BLK_LEN, SK_BLK_DETAIL = 1, 2, 4, 8
SK_DETAIL_FULL, SK_DETAIL_MATCHED = 0, 1
SK_TABLE_NONE, SK_TABLE_FULL_KEY, SK_TABLE_FACTORED, SK_TABLE_WIDE_CLASSES = 0, 1, 2, 4


class BlockedLayout(C.Structure):
    """sk_blocked_layout: where each array of a 64-cluster tile sits inside the tile's input / output block."""
    _fields_ = [("n_mates", C.c_int32), ("stride", C.c_int32), ("bc_stride", C.c_int32), ("flags", C.c_int32),
                ("in_block", C.c_int32), ("out_block", C.c_int32),
                ("in_qual", C.c_int32 * 2), ("in_seq", C.c_int32 * 2), ("in_len", C.c_int32 * 2), ("in_bc", C.c_int32),
                ("out_seq", C.c_int32 * 2), ("out_lowest_k", C.c_int32 * 2), ("out_assign", C.c_int32),
                ("out_lowest_diff", C.c_int32), ("out_first_idx", C.c_int32), ("out_last_idx", C.c_int32)]

    def ntiles(self, n: int) -> int:
        return (n + 63) // 64

    def in_bytes(self, n: int) -> int:
        return self.ntiles(n) * self.in_block

    def out_bytes(self, n: int) -> int:
        return self.ntiles(n) * self.out_block

    # ---- host-side packer / unpacker (numpy views only: bytes are moved, nothing is computed) ----------------
    def _seg(self, buf: np.ndarray, block: int, off: int, row_bytes: int) -> np.ndarray:
        """[ntiles, 64, row_bytes] view of one segment of every block."""
        nt = buf.size // block
        return np.lib.stride_tricks.as_strided(buf[off:], shape=(nt, 64, row_bytes), strides=(block, row_bytes, 1))

    def pack(self, mates, bc=None) -> np.ndarray:
        """mates: list of (seq, qual, len-or-None) uint8 [n, stride] matrices -> the input buffer (uint8, whole blocks)."""
        n = mates[0][1].shape[0]
        nt = self.ntiles(n)
        buf = np.zeros(nt * self.in_block, dtype=np.uint8)

        def put(off, mat, row_bytes):
            seg = self._seg(buf, self.in_block, off, row_bytes)
            full = n // 64
            m2 = np.ascontiguousarray(mat).view(np.uint8).reshape(n, row_bytes)
            if full:
                seg[:full] = m2[:full * 64].reshape(full, 64, row_bytes)
            if n % 64:
                seg[full, :n % 64] = m2[full * 64:]
        for m, (seq, qual, length) in enumerate(mates):
            put(self.in_qual[m], qual, self.stride)
            if self.in_seq[m] >= 0:
                put(self.in_seq[m], seq, self.stride)
            if self.in_len[m] >= 0:
                put(self.in_len[m], np.asarray(length, dtype=np.uint16), 2)
        if self.in_bc >= 0:
            put(self.in_bc, bc, self.bc_stride)
        return buf

    def unpack(self, out: np.ndarray, n: int) -> dict:
        def get(off, row_bytes, dtype=None):
            seg = self._seg(out, self.out_block, off, row_bytes)
            flat = np.ascontiguousarray(seg).reshape(-1, row_bytes)[:n]
            return flat if dtype is None else np.ascontiguousarray(flat).view(dtype).reshape(n)
        res = {"out_seq": [], "lowest_k": []}
        for m in range(self.n_mates):
            if self.out_seq[m] >= 0:
                res["out_seq"].append(get(self.out_seq[m], self.stride))
            if self.out_lowest_k[m] >= 0:
                res["lowest_k"].append(get(self.out_lowest_k[m], 2, np.uint16))
        if self.out_assign >= 0:
            res["assign"] = get(self.out_assign, 4, np.int32)
        if self.out_lowest_diff >= 0:
            res["lowest_diff"] = get(self.out_lowest_diff, 1, np.uint8)
            res["first_idx"] = get(self.out_first_idx, 2, np.int16)
            res["last_idx"] = get(self.out_last_idx, 2, np.int16)
        return res


_libs: dict = {}


def library_path() -> str:
    return _build.LIB_PATH


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libseqkit_hip.so (no GPU needed for this step) and declare the prototypes.
    `path` selects another build of the same C-ABI (A/B timing of kernel variants in one process)."""
    path = path or library_path()
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise SeqkitHipError(
            f"{path} is missing: build it with `python -m seqkit_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for this library.")
    lib = C.CDLL(path)
    vp, i32, i64, u8 = C.c_void_p, C.c_int, C.c_int64, C.c_uint8
    protos = {
        "sk_version": (i32, []), "sk_device_count": (i32, []),
        "sk_create": (i32, [i32, C.POINTER(vp)]), "sk_destroy": (None, [vp]),
        "sk_last_error": (C.c_char_p, [vp]), "sk_sync": (i32, [vp]), "sk_stream": (vp, [vp]),
        "sk_malloc_device": (i32, [vp, C.c_size_t, C.POINTER(vp)]), "sk_free_device": (i32, [vp, vp]),
        "sk_malloc_pinned": (i32, [vp, C.c_size_t, C.POINTER(vp)]), "sk_free_pinned": (i32, [vp, vp]),
        "sk_copy_h2d": (i32, [vp, vp, vp, C.c_size_t]), "sk_copy_d2h": (i32, [vp, vp, vp, C.c_size_t]),
        "sk_set_barcodes": (i32, [vp, vp, i32, i32, i32]), "sk_set_detail_mode": (i32, [vp, i32]),
        "sk_barcode_table_info": (i32, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "sk_demux_assign": (i32, [vp, vp, i32, i64, vp, vp, vp, vp]),
        "sk_demux_assign_dev": (i32, [vp, vp, i32, i64, vp, vp, vp, vp, vp]),
        "sk_trim_by_quality": (i32, [vp, vp, vp, i32, i64, u8, vp]),
        "sk_trim_by_quality_dev": (i32, [vp, vp, vp, i32, i64, u8, vp]),
        "sk_mask_by_quality": (i32, [vp, vp, vp, vp, i32, i64, u8]),
        "sk_mask_by_quality_dev": (i32, [vp, vp, vp, i32, i64, u8, vp]),
        "sk_fused_pass": (i32, [vp, C.POINTER(_FusedArgs)]),
        "sk_fused_pass_dev": (i32, [vp, C.POINTER(_FusedArgs)]),
        "sk_fused_pass_many_dev": (i32, [vp, C.POINTER(_FusedArgs), i32]),
        "sk_demux_assign_many_dev": (i32, [vp, C.POINTER(_DemuxBatch), i32, i32]),
        "sk_trim_by_quality_many_dev": (i32, [vp, C.POINTER(_TrimBatch), i32, i32, u8]),
        "sk_blocked_layout_init": (i32, [C.POINTER(BlockedLayout), i32, i32, i32, i32]),
        "sk_fused_pass_blocked_dev": (i32, [vp, C.POINTER(BlockedLayout), vp, vp, i64, u8, vp]),
        "sk_fused_tune_placement_dev": (i32, [vp, C.POINTER(_FusedArgs), C.POINTER(_FusedCandidates), i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                              C.POINTER(i32)]),
        "sk_counts_reset": (i32, [vp]), "sk_counts_get": (i32, [vp, vp]), "sk_counts_device_ptr": (vp, [vp]),
        "sk_comm_ready": (i32, [vp]), "sk_comm_get_unique_id": (i32, [vp]), "sk_comm_init_rank": (i32, [vp, vp, i32, i32]), "sk_comm_destroy": (i32, [vp]),
        "sk_counts_allreduce": (i32, [C.POINTER(vp), i32]), "sk_allreduce_u64_dev": (i32, [vp, vp, C.c_size_t]),
        "sk_bam_flag_tlen": (i32, [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
        "sk_bam_flag_tlen_dev": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
        "sk_bgzf_deflate": (i32, [vp, vp, C.c_size_t, vp, i64, vp, C.c_size_t, vp]),
        "sk_bgzf_deflate_dev": (i32, [vp, vp, vp, i64, vp, C.c_uint32, vp, vp, vp]),
        "sk_bgzf_inflate_dev": (i32, [vp, vp, vp, i64, vp, vp, i32]),
        "sk_bam_walk_dev": (i32, [vp, vp, C.c_uint64, vp, i64, C.c_uint64, i32, vp, vp, vp, i32, C.POINTER(i32), C.POINTER(C.c_uint64), C.POINTER(i32)]),
        "sk_bam_walk_reduce_dev": (i32, [vp, vp, C.c_uint64, vp, vp, i64, i32, i32, i32, vp]),
        "sk_bam_file_reduce": (i32, [vp, C.c_char_p, i32, vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_double)]),
        "sk_bam_fragments": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, vp, vp]),
        "sk_bam_fragments_dev": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, vp, vp]),
        "sk_bam_sequence": (i32, [vp, vp, i32, vp, i32, vp, vp, i64, C.c_uint8, vp]),
        "sk_bam_sequence_dev": (i32, [vp, vp, i32, vp, i32, vp, vp, i64, C.c_uint8, vp]),
        "sk_count_set_regions": (i32, [vp, i32, vp, vp, vp, vp, i64, i64]),
        "sk_count_add": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, C.c_uint8, C.c_uint32, i32, i32]),
        "sk_count_add_dev": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, C.c_uint8, C.c_uint32, i32, i32]),
        "sk_count_get": (i32, [vp, vp]),
        "sk_gc_set_genome": (i32, [vp, vp, i64]),
        "sk_gc_count": (i32, [vp, vp, vp, i64, vp, vp]),
        "sk_census_reset": (i32, [vp]),
        "sk_census_add": (i32, [vp, vp, i32, i32, i64, vp, i64]),
        "sk_census_add_dev": (i32, [vp, vp, i32, i32, i64, vp, i64]),
        "sk_census_stats": (i32, [vp, vp]),
        "sk_census_count_hist": (i32, [vp, vp]),
        "sk_census_entries": (i32, [vp, C.c_uint64, vp, C.c_uint64, vp]),
        "sk_timer_start": (i32, [vp]), "sk_timer_stop": (i32, [vp, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in protos.items():
        if path != library_path() and not hasattr(lib, name):
            continue                                    # an older build of the C-ABI loaded for A/B timing: what it has is enough
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def blocked_layout(n_mates: int, stride: int, bc_stride: int, flags: int, lib=None) -> BlockedLayout:
    lay = BlockedLayout()
    rc = (lib or load_library()).sk_blocked_layout_init(C.byref(lay), n_mates, stride, bc_stride, flags)
    if rc != 0:
        raise SeqkitHipError(f"sk_blocked_layout_init({n_mates}, {stride}, {bc_stride}, {flags}) failed ({rc})")
    return lay


def comm_unique_id(lib=None) -> bytes:
    """sk_comm_get_unique_id: 128 opaque bytes rank 0 hands to every rank (ncclGetUniqueId underneath)."""
    buf = (C.c_uint8 * 128)()
    rc = (lib or load_library()).sk_comm_get_unique_id(buf)
    if rc != 0:
        raise SeqkitHipError(f"sk_comm_get_unique_id failed ({rc})")
    return bytes(buf)


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


def _mat(a, name: str) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2 or not a.flags.c_contiguous:
        raise ValueError(f"{name} must be a C-contiguous uint8 matrix [n, stride]")
    return a


def _vec(a, dtype, n: int, name: str) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != dtype or a.ndim != 1 or a.shape[0] != n or not a.flags.c_contiguous:
        raise ValueError(f"{name} must be a contiguous {np.dtype(dtype).name} vector of length {n}")
    return a


class Context:
    """One sk_ctx: one GPU, one stream.  Mirrors the C-ABI call for call."""

    def __init__(self, device: int = 0, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        h = C.c_void_p()
        rc = self._lib.sk_create(device, C.byref(h))
        if rc != 0:
            msg = self._lib.sk_last_error(None).decode()
            raise SeqkitHipError(f"sk_create({device}) failed ({rc}): {msg}")
        self._h = h
        self.S = 0
        self.L = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.sk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise SeqkitHipError(f"{what} failed ({rc}): {self._lib.sk_last_error(self._h).decode()}")

    # ---- lifetime / memory --------------------------------------------------------------
    def sync(self) -> None:
        self._check(self._lib.sk_sync(self._h), "sk_sync")

    # ---- F2 on the device: BGZF deflate (host pointers) ----
    DEFLATE_BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("reserved", "<u4")])

    def bgzf_deflate(self, data: bytes, block: int = 0xff00) -> bytes:
        """sk_bgzf_deflate: `data` cut into blocks of `block` bytes, every one a BGZF member; the members back to back."""
        n = max(1, -(-len(data) // block)) if data else 0
        blocks = np.zeros(n, dtype=self.DEFLATE_BLOCK_DTYPE)
        for i in range(n):
            blocks[i] = (i * block, min(block, len(data) - i * block), 0)
        src = np.frombuffer(data + bytes(8), dtype=np.uint8)
        out = np.empty(n * 65536 + 64, dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        self._check(self._lib.sk_bgzf_deflate(self._h, src.ctypes.data, len(data), blocks.ctypes.data, n, out.ctypes.data, out.nbytes, off.ctypes.data), "sk_bgzf_deflate")
        return out[:int(off[n])].tobytes()

    # ---- B1 on the device: BGZF inflate, record walk (all pointers are device addresses) ----
    BGZF_BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("out_off", "<u8"), ("crc32", "<u4"), ("reserved", "<u4")])

    def bgzf_inflate_dev(self, comp: int, blocks: int, n_blocks: int, out: int, status: int, check_crc: bool = True) -> None:
        self._check(self._lib.sk_bgzf_inflate_dev(self._h, comp, blocks, n_blocks, out, status, 1 if check_crc else 0), "sk_bgzf_inflate_dev")

    def bam_walk_dev(self, stream: int, stream_len: int, block_end: int, n: int, first_record: int, entry: int, exit_scratch: int, nrec_scratch: int,
                     max_rounds: int = 64, n_ref: int = -1):
        ver, nrec, rounds = C.c_int32(0), C.c_uint64(0), C.c_int32(0)
        self._check(self._lib.sk_bam_walk_dev(self._h, stream, stream_len, block_end, n, first_record, n_ref, entry, exit_scratch, nrec_scratch, max_rounds,
                                              C.byref(ver), C.byref(nrec), C.byref(rounds)), "sk_bam_walk_dev")
        return bool(ver.value), int(nrec.value), int(rounds.value)

    def bam_walk_reduce_dev(self, stream: int, stream_len: int, block_end: int, entry: int, n: int, max_frag: int, out: int,
                            want_counters: bool = True, want_hist: bool = True) -> None:
        self._check(self._lib.sk_bam_walk_reduce_dev(self._h, stream, stream_len, block_end, entry, n, max_frag, 1 if want_counters else 0,
                                                     1 if want_hist else 0, out), "sk_bam_walk_reduce_dev")

    def bam_file_reduce(self, path: str, max_frag: int = 5000, want_counters: bool = True, want_hist: bool = True):
        """sk_bam_file_reduce: (handled, counters u64[3], hist u64[max_frag + 1], hist_total, info f64[8])."""
        counters = np.zeros(3, dtype=np.uint64)
        hist = np.zeros(max_frag + 1, dtype=np.uint64)
        total = np.zeros(1, dtype=np.uint64)
        handled = C.c_int32(0)
        info = (C.c_double * 8)()
        self._check(self._lib.sk_bam_file_reduce(self._h, os.fsencode(path), max_frag, counters.ctypes.data if want_counters else None,
                                                 hist.ctypes.data if want_hist else None, total.ctypes.data, C.byref(handled), info), "sk_bam_file_reduce")
        return bool(handled.value), counters, hist, int(total[0]), [float(x) for x in info]

    def stream(self) -> int:
        return int(self._lib.sk_stream(self._h) or 0)

    def malloc_device(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.sk_malloc_device(self._h, nbytes, C.byref(p)), "sk_malloc_device")
        return int(p.value)

    def free_device(self, p: int) -> None:
        self._check(self._lib.sk_free_device(self._h, p), "sk_free_device")

    def pinned_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array in page-locked host memory (sk_malloc_pinned): the host entry points move such buffers by DMA.
        The memory belongs to the ctx's process until free_pinned(array)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.sk_malloc_pinned(self._h, max(nbytes, 1), C.byref(p)), "sk_malloc_pinned")
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def free_pinned(self, arr: np.ndarray) -> None:
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self._lib.sk_free_pinned(self._h, p), "sk_free_pinned")

    def copy_h2d(self, dst: int, src: np.ndarray) -> None:
        self._check(self._lib.sk_copy_h2d(self._h, dst, src.ctypes.data, src.nbytes), "sk_copy_h2d")

    def copy_d2h(self, dst: np.ndarray, src: int) -> None:
        self._check(self._lib.sk_copy_d2h(self._h, dst.ctypes.data, src, dst.nbytes), "sk_copy_d2h")

    def timer_start(self) -> None:
        self._check(self._lib.sk_timer_start(self._h), "sk_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._lib.sk_timer_stop(self._h, C.byref(ms)), "sk_timer_stop")
        return float(ms.value)

    # ---- barcodes -----------------------------------------------------------------------
    def set_barcodes(self, table, max_diff: int = 1) -> None:
        table = np.ascontiguousarray(table, dtype=np.uint8)
        if table.ndim != 2:
            raise ValueError("table must be [S, L] uint8")
        S, L = table.shape
        self._check(self._lib.sk_set_barcodes(self._h, _ptr(table) if S else None, S, L, max_diff), "sk_set_barcodes")
        self.S, self.L = S, L

    def set_detail_mode(self, mode: int) -> None:
        """DETAIL_FULL: lowest_diff/first/last of every row; DETAIL_MATCHED: of rows with assign != -1 only (include/seqkit_hip.h)."""
        self._check(self._lib.sk_set_detail_mode(self._h, mode), "sk_set_detail_mode")

    def barcode_table_info(self) -> dict:
        """What serves demultiplex-alone calls of the current sheet: kind (TABLE_NONE / TABLE_FULL_KEY / TABLE_FACTORED, | TABLE_WIDE_CLASSES), keys, bytes."""
        kind, keys, nbytes = C.c_int(), C.c_int64(), C.c_int64()
        self._check(self._lib.sk_barcode_table_info(self._h, C.byref(kind), C.byref(keys), C.byref(nbytes)), "sk_barcode_table_info")
        return {"kind": int(kind.value), "keys": int(keys.value), "bytes": int(nbytes.value)}

    def counts_reset(self) -> None:
        self._check(self._lib.sk_counts_reset(self._h), "sk_counts_reset")

    def counts(self) -> np.ndarray:
        out = np.zeros(self.S + 3, dtype=np.uint64)
        self._check(self._lib.sk_counts_get(self._h, _ptr(out)), "sk_counts_get")
        return out

    def counts_device_ptr(self) -> int:
        return int(self._lib.sk_counts_device_ptr(self._h) or 0)

    # ---- (e) the count reduce over RCCL ----------------------------------------------------
    def comm_init_rank(self, unique_id: bytes, rank: int, n_ranks: int) -> None:
        """One process per GPU: join the communicator whose 128-byte id rank 0 made with comm_unique_id()."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.sk_comm_init_rank(self._h, buf, rank, n_ranks), "sk_comm_init_rank")

    def comm_ready(self) -> None:
        """Rank-local: raises unless this ctx could join an RCCL communicator (library loadable, device bindable)."""
        self._check(self._lib.sk_comm_ready(self._h), "sk_comm_ready")

    def comm_destroy(self) -> None:
        self._check(self._lib.sk_comm_destroy(self._h), "sk_comm_destroy")

    def counts_allreduce(self, others=()) -> None:
        """Sum the u64[S+3] counters over `self` + `others` (one process, several ctxs), or — alone, after
        comm_init_rank — over the ranks."""
        ctxs = [self] + list(others)
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        self._check(self._lib.sk_counts_allreduce(arr, len(ctxs)), "sk_counts_allreduce")

    def allreduce_u64_dev(self, buf: int, count: int) -> None:
        self._check(self._lib.sk_allreduce_u64_dev(self._h, buf, count), "sk_allreduce_u64_dev")

    # ---- host entry points --------------------------------------------------------------
    def demux_assign(self, bc, want_detail: bool = True):
        bc = _mat(bc, "bc")
        n, bstride = bc.shape
        assign = np.empty(n, dtype=np.int32)
        low = np.empty(n, dtype=np.uint8) if want_detail else None
        first = np.empty(n, dtype=np.int16) if want_detail else None
        last = np.empty(n, dtype=np.int16) if want_detail else None
        self._check(self._lib.sk_demux_assign(self._h, _ptr(bc), bstride, n, _ptr(assign), _ptr(low), _ptr(first),
                                              _ptr(last)), "sk_demux_assign")
        return assign, low, first, last

    def trim_by_quality(self, qual, length, min_baseq: int) -> np.ndarray:
        qual = _mat(qual, "qual")
        n, stride = qual.shape
        length = None if length is None else _vec(length, np.uint16, n, "len")
        out = np.empty(n, dtype=np.uint16)
        self._check(self._lib.sk_trim_by_quality(self._h, _ptr(qual), _ptr(length), stride, n, min_baseq, _ptr(out)),
                    "sk_trim_by_quality")
        return out

    def mask_by_quality(self, seq, qual, length, min_baseq: int) -> np.ndarray:
        """Returns the masked copy (the C entry point works in place on its seq argument)."""
        seq = _mat(seq, "seq").copy()
        qual = _mat(qual, "qual")
        if seq.shape != qual.shape:
            raise ValueError("seq and qual must have the same shape")
        n, stride = seq.shape
        length = None if length is None else _vec(length, np.uint16, n, "len")
        self._check(self._lib.sk_mask_by_quality(self._h, _ptr(seq), _ptr(qual), _ptr(length), stride, n, min_baseq),
                    "sk_mask_by_quality")
        return seq

    def fused_pass(self, mates, min_baseq: int, bc=None, want_detail: bool = False, do_mask: bool = True,
                   do_trim: bool = True):
        """mates: list of (seq, qual, len-or-None).  Returns dict with per-mate outputs and the assignment."""
        a = _FusedArgs()
        keep = []
        n = None
        stride = 0
        res = {"out_seq": [], "lowest_k": []}
        a.n_mates = len(mates)
        for i, (seq, qual, length) in enumerate(mates):
            qual = _mat(qual, "qual")
            if n is None:
                n, stride = qual.shape
            if qual.shape != (n, stride):
                raise ValueError("all mates must share [n, stride]")
            a.mate[i].qual = _ptr(qual)
            keep.append(qual)
            if length is not None:
                length = _vec(length, np.uint16, n, "len")
                a.mate[i].len = _ptr(length)
                keep.append(length)
            if do_mask:
                seq = _mat(seq, "seq")
                out = np.empty_like(seq)
                a.mate[i].seq = _ptr(seq)
                a.mate[i].out_seq = _ptr(out)
                keep.append(seq)
                res["out_seq"].append(out)
            if do_trim:
                lk = np.empty(n, dtype=np.uint16)
                a.mate[i].lowest_k = _ptr(lk)
                res["lowest_k"].append(lk)
        if bc is not None:
            bc = _mat(bc, "bc")
            if n is None:
                n = bc.shape[0]
            a.bc = _ptr(bc)
            a.bc_stride = bc.shape[1]
            res["assign"] = np.empty(n, dtype=np.int32)
            a.assign = _ptr(res["assign"])
            if want_detail:
                res["lowest_diff"] = np.empty(n, dtype=np.uint8)
                res["first_idx"] = np.empty(n, dtype=np.int16)
                res["last_idx"] = np.empty(n, dtype=np.int16)
                a.lowest_diff = _ptr(res["lowest_diff"])
                a.first_idx = _ptr(res["first_idx"])
                a.last_idx = _ptr(res["last_idx"])
        a.n = n or 0
        a.stride = stride
        a.min_baseq = min_baseq
        self._check(self._lib.sk_fused_pass(self._h, C.byref(a)), "sk_fused_pass")
        return res

    def bam_flag_tlen(self, flag, tid, mtid, tlen, max_frag: int = 5000):
        n = len(flag)
        flag = _vec(flag, np.uint16, n, "flag")
        tid = _vec(tid, np.int32, n, "tid")
        mtid = _vec(mtid, np.int32, n, "mtid")
        tlen = _vec(tlen, np.int32, n, "tlen")
        counters = np.zeros(3, dtype=np.uint64)
        hist = np.zeros(max_frag + 1, dtype=np.uint64)
        total = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sk_bam_flag_tlen(self._h, _ptr(flag), _ptr(tid), _ptr(mtid), _ptr(tlen), n, max_frag,
                                               _ptr(counters), _ptr(hist), _ptr(total)), "sk_bam_flag_tlen")
        return counters, hist, int(total[0])

    def bam_fragments(self, flag, tid, mtid, tlen, min_size: int = 0, max_size: int = 5000):
        """keep[n] (0/1) and the number kept: src/sam_fragments.rs:27-38."""
        n = len(flag)
        flag = _vec(flag, np.uint16, n, "flag")
        tid = _vec(tid, np.int32, n, "tid")
        mtid = _vec(mtid, np.int32, n, "mtid")
        tlen = _vec(tlen, np.int32, n, "tlen")
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        kept = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sk_bam_fragments(self._h, _ptr(flag), _ptr(tid), _ptr(mtid), _ptr(tlen), n, min_size, max_size,
                                               _ptr(bits), _ptr(kept)), "sk_bam_fragments")
        return np.unpackbits(bits, bitorder="little")[:n], int(kept[0])

    def bam_fragments_dev(self, flag: int, tid: int, mtid: int, tlen: int, n: int, min_size: int, max_size: int, keep_bits: int,
                          kept: int) -> None:
        self._check(self._lib.sk_bam_fragments_dev(self._h, flag, tid, mtid, tlen, n, min_size, max_size, keep_bits, kept),
                    "sk_bam_fragments_dev")

    # ---- f2 (second half): sam count -------------------------------------------------------
    def count_set_regions(self, chr_off, rstart, rend, ridx=None, n_regions: int | None = None) -> None:
        """Regions grouped by BAM reference: entries chr_off[c]..chr_off[c+1]-1 belong to reference c; ridx maps an entry
        to its index in the caller's list of n_regions regions."""
        chr_off = np.ascontiguousarray(chr_off, dtype=np.int32)
        rstart = np.ascontiguousarray(rstart, dtype=np.uint32)
        rend = np.ascontiguousarray(rend, dtype=np.uint32)
        n = len(rstart)
        ridx = None if ridx is None else np.ascontiguousarray(ridx, dtype=np.int32)
        self._n_regions = n if n_regions is None else n_regions
        self._check(self._lib.sk_count_set_regions(self._h, len(chr_off) - 1, _ptr(chr_off), _ptr(rstart), _ptr(rend), _ptr(ridx), n,
                                                   self._n_regions), "sk_count_set_regions")

    def count_add(self, flag, mapq, tid, mtid, pos, mpos, tlen, end_pos=None, min_mapq: int = 0, max_frag_len: int = 5000,
                  single_end: bool = False, center: bool = False) -> None:
        n = len(flag)
        cols = [_vec(flag, np.uint16, n, "flag"), _vec(mapq, np.uint8, n, "mapq")]
        for name, col in (("tid", tid), ("mtid", mtid), ("pos", pos), ("mpos", mpos), ("tlen", tlen), ("end_pos", end_pos)):
            cols.append(None if col is None else _vec(col, np.int32, n, name))
        self._check(self._lib.sk_count_add(self._h, *[_ptr(c) for c in cols], n, min_mapq, max_frag_len, int(single_end), int(center)),
                    "sk_count_add")

    def count_add_dev(self, flag: int, mapq: int, tid: int, mtid: int, pos: int, mpos: int, tlen: int, end_pos: int, n: int,
                      min_mapq: int = 0, max_frag_len: int = 5000, single_end: bool = False, center: bool = False) -> None:
        self._check(self._lib.sk_count_add_dev(self._h, flag, mapq, tid, mtid, pos, mpos, tlen, end_pos or None, n, min_mapq, max_frag_len,
                                               int(single_end), int(center)), "sk_count_add_dev")

    def count_get(self) -> np.ndarray:
        out = np.zeros(max(self._n_regions, 1), dtype=np.uint32)
        self._check(self._lib.sk_count_get(self._h, _ptr(out)), "sk_count_get")
        return out[:self._n_regions]

    # ---- fasta gc content -------------------------------------------------------------------
    def gc_set_genome(self, genome) -> None:
        g = np.ascontiguousarray(np.frombuffer(genome, dtype=np.uint8) if isinstance(genome, (bytes, bytearray)) else genome, dtype=np.uint8)
        self._check(self._lib.sk_gc_set_genome(self._h, _ptr(g) if g.size else None, g.size), "sk_gc_set_genome")

    def gc_count(self, start, length):
        start = np.ascontiguousarray(start, dtype=np.int64)
        length = np.ascontiguousarray(length, dtype=np.int64)
        n = len(start)
        gc = np.zeros(max(n, 1), dtype=np.uint64)
        total = np.zeros(max(n, 1), dtype=np.uint64)
        self._check(self._lib.sk_gc_count(self._h, _ptr(start), _ptr(length), n, _ptr(gc), _ptr(total)), "sk_gc_count")
        return gc[:n], total[:n]

    # ---- f4: sam to fastq sequence() ---------------------------------------------------------
    def bam_sequence(self, seq4, qual, length, flag, min_baseq: int = 10) -> np.ndarray:
        """ASCII bases [n, stride]: src/sam_to_fastq.rs:31-59 (bytes past a row's length are unspecified)."""
        seq4 = _mat(seq4, "seq4")
        qual = _mat(qual, "qual")
        n, stride = qual.shape
        if seq4.shape[0] != n:
            raise ValueError("seq4 and qual must have the same number of rows")
        ln = None if length is None else _vec(length, np.uint16, n, "length")
        flag = _vec(flag, np.uint16, n, "flag")
        out = np.zeros((n, stride), dtype=np.uint8)
        self._check(self._lib.sk_bam_sequence(self._h, _ptr(seq4), seq4.shape[1], _ptr(qual), stride, _ptr(ln), _ptr(flag), n, min_baseq,
                                              _ptr(out)), "sk_bam_sequence")
        return out

    def bam_sequence_dev(self, seq4: int, seq4_stride: int, qual: int, stride: int, length: int, flag: int, n: int, min_baseq: int,
                         out: int) -> None:
        self._check(self._lib.sk_bam_sequence_dev(self._h, seq4, seq4_stride, qual, stride, length or None, flag, n, min_baseq, out),
                    "sk_bam_sequence_dev")

    # ---- f3: barcode census ----------------------------------------------------------------
    def census_reset(self) -> None:
        self._check(self._lib.sk_census_reset(self._h), "sk_census_reset")

    def census_add(self, bc, L: int | None = None, assign=None, row_base: int = 0) -> None:
        """bc: uint8 [n, bc_stride] host matrix (rows NUL-padded); assign: optional int32[n]."""
        bc = np.ascontiguousarray(bc, dtype=np.uint8)
        n, stride = bc.shape
        if assign is not None:
            assign = _vec(assign, np.int32, n, "assign")
        self._check(self._lib.sk_census_add(self._h, _ptr(bc), stride, stride if L is None else L, n,
                                            _ptr(assign) if assign is not None else None, row_base), "sk_census_add")

    def census_add_dev(self, bc: int, bc_stride: int, L: int, n: int, assign: int = 0, row_base: int = 0) -> None:
        self._check(self._lib.sk_census_add_dev(self._h, bc, bc_stride, L, n, assign or None, row_base), "sk_census_add_dev")

    def census_stats(self) -> dict:
        s = np.zeros(4, dtype=np.uint64)
        self._check(self._lib.sk_census_stats(self._h, _ptr(s)), "sk_census_stats")
        return {"distinct": int(s[0]), "counted": int(s[1]), "rejected": int(s[2]), "slots": int(s[3])}

    def census_count_hist(self):
        h = np.zeros(64, dtype=np.uint64)
        self._check(self._lib.sk_census_count_hist(self._h, _ptr(h)), "sk_census_count_hist")
        return h

    def census_entries(self, min_count: int = 1, cap: int | None = None):
        """[(barcode bytes, count, first_row)] in first-seen order, and how many barcodes qualified in all."""
        if cap is None:
            cap = self.census_stats()["distinct"]
        dt = np.dtype([("barcode", "S32"), ("count", np.uint64), ("first_row", np.int64)])
        out = np.zeros(max(cap, 1), dtype=dt)
        total = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.sk_census_entries(self._h, min_count, _ptr(out), cap, _ptr(total)), "sk_census_entries")
        k = min(cap, int(total[0]))
        return [(bytes(e["barcode"]), int(e["count"]), int(e["first_row"])) for e in out[:k]], int(total[0])

    # ---- device entry points (raw addresses) ---------------------------------------------
    def fused_pass_dev(self, n: int, stride: int, min_baseq: int, mates, bc: int = 0, bc_stride: int = 0,
                       assign: int = 0, lowest_diff: int = 0, first_idx: int = 0, last_idx: int = 0,
                       counts: int = 0) -> None:
        """mates: list of dicts with device addresses {seq, qual, len, out_seq, lowest_k} (0 = NULL)."""
        a = _FusedArgs()
        a.n, a.n_mates, a.stride, a.min_baseq = n, len(mates), stride, min_baseq
        for i, m in enumerate(mates):
            a.mate[i].seq = m.get("seq") or None
            a.mate[i].qual = m.get("qual") or None
            a.mate[i].len = m.get("len") or None
            a.mate[i].out_seq = m.get("out_seq") or None
            a.mate[i].lowest_k = m.get("lowest_k") or None
        a.bc = bc or None
        a.bc_stride = bc_stride
        a.assign = assign or None
        a.lowest_diff = lowest_diff or None
        a.first_idx = first_idx or None
        a.last_idx = last_idx or None
        a.counts = counts or None
        self._check(self._lib.sk_fused_pass_dev(self._h, C.byref(a)), "sk_fused_pass_dev")

    def fused_pass_blocked_dev(self, lay: BlockedLayout, inp: int, out: int, n: int, min_baseq: int, counts: int = 0) -> None:
        self._check(self._lib.sk_fused_pass_blocked_dev(self._h, C.byref(lay), inp, out, n, min_baseq, counts or None),
                    "sk_fused_pass_blocked_dev")

    def fused_pass_blocked(self, mates, min_baseq: int, bc=None, want_detail: bool = False, do_mask: bool = True,
                           do_trim: bool = True) -> dict:
        """The tile-blocked pass on host matrices: pack -> device -> sk_fused_pass_blocked_dev -> unpack (tests)."""
        n, stride = mates[0][1].shape
        ragged = mates[0][2] is not None
        flags = (SK_BLK_MASK if do_mask else 0) | (SK_BLK_TRIM if do_trim else 0) | (SK_BLK_LEN if ragged else 0) | \
                (SK_BLK_DETAIL if want_detail else 0)
        lay = blocked_layout(len(mates), stride, 0 if bc is None else bc.shape[1], flags, self._lib)
        hin = lay.pack(mates, bc)
        hout = np.empty(lay.out_bytes(n), dtype=np.uint8)
        din, dout = self.malloc_device(hin.nbytes + 16), self.malloc_device(hout.nbytes + 16)
        try:
            self.copy_h2d(din, hin)
            self.fused_pass_blocked_dev(lay, din, dout, n, min_baseq)
            self.copy_d2h(hout, dout)
            self.sync()
        finally:
            self.free_device(din)
            self.free_device(dout)
        return lay.unpack(hout, n)

    def fused_tune_placement_dev(self, n: int, stride: int, min_baseq: int, mates, cands, bc: int = 0, bc_stride: int = 0, assign: int = 0,
                                 sweeps: int = 2):
        """mates as in fused_pass_dev; cands: list per mate of {"seq": [ptr...], "qual": [...], "out_seq": [...]} (same k everywhere).
        Returns (mates with the chosen pointers, ms_before, ms_after, n_probes)."""
        a = _FusedArgs()
        a.n, a.n_mates, a.stride, a.min_baseq = n, len(mates), stride, min_baseq
        for i, m in enumerate(mates):
            a.mate[i].seq = m.get("seq") or None
            a.mate[i].qual = m.get("qual") or None
            a.mate[i].len = m.get("len") or None
            a.mate[i].out_seq = m.get("out_seq") or None
            a.mate[i].lowest_k = m.get("lowest_k") or None
        a.bc, a.bc_stride, a.assign = bc or None, bc_stride, assign or None
        cd = _FusedCandidates()
        cd.k = len(cands[0]["qual"])
        for i, cm in enumerate(cands):
            for k in range(cd.k):
                cd.seq[i][k] = cm["seq"][k] if cm.get("seq") else None
                cd.qual[i][k] = cm["qual"][k]
                cd.out_seq[i][k] = cm["out_seq"][k] if cm.get("out_seq") else None
        before, after, probes = C.c_float(), C.c_float(), C.c_int()
        self._check(self._lib.sk_fused_tune_placement_dev(self._h, C.byref(a), C.byref(cd), sweeps, C.byref(before), C.byref(after), C.byref(probes)),
                    "sk_fused_tune_placement_dev")
        chosen = [dict(m, seq=a.mate[i].seq or 0, qual=a.mate[i].qual or 0, out_seq=a.mate[i].out_seq or 0) for i, m in enumerate(mates)]
        return chosen, float(before.value), float(after.value), int(probes.value)

    def demux_assign_dev(self, bc: int, bc_stride: int, n: int, assign: int, lowest_diff: int = 0, first_idx: int = 0,
                         last_idx: int = 0, counts: int = 0) -> None:
        self._check(self._lib.sk_demux_assign_dev(self._h, bc, bc_stride, n, assign, lowest_diff or None,
                                                  first_idx or None, last_idx or None, counts or None),
                    "sk_demux_assign_dev")

    def trim_by_quality_dev(self, qual: int, length: int, stride: int, n: int, min_baseq: int, lowest_k: int) -> None:
        self._check(self._lib.sk_trim_by_quality_dev(self._h, qual, length or None, stride, n, min_baseq, lowest_k),
                    "sk_trim_by_quality_dev")

    def mask_by_quality_dev(self, seq: int, qual: int, stride: int, n: int, min_baseq: int, out_seq: int) -> None:
        self._check(self._lib.sk_mask_by_quality_dev(self._h, seq, qual, stride, n, min_baseq, out_seq),
                    "sk_mask_by_quality_dev")

    def demux_assign_many_dev(self, batches, bc_stride: int) -> None:
        """batches: [(bc, n, assign, lowest_diff or 0, first_idx or 0, last_idx or 0)] of device addresses: sk_demux_assign_many_dev"""
        arr = (_DemuxBatch * len(batches))()
        for i, b in enumerate(batches):
            bc, n, assign, low, first, last = (list(b) + [0, 0, 0])[:6]
            arr[i] = _DemuxBatch(bc, n, assign, low or None, first or None, last or None)
        self._check(self._lib.sk_demux_assign_many_dev(self._h, arr, len(batches), bc_stride), "sk_demux_assign_many_dev")

    def trim_by_quality_many_dev(self, batches, stride: int, min_baseq: int) -> None:
        """batches: [(qual, len or 0, n, lowest_k)] of device addresses: sk_trim_by_quality_many_dev"""
        arr = (_TrimBatch * len(batches))()
        for i, (qual, ln, n, lowest_k) in enumerate(batches):
            arr[i] = _TrimBatch(qual, ln or None, n, lowest_k)
        self._check(self._lib.sk_trim_by_quality_many_dev(self._h, arr, len(batches), stride, min_baseq), "sk_trim_by_quality_many_dev")

    def bam_flag_tlen_dev(self, flag: int, tid: int, mtid: int, tlen: int, n: int, max_frag: int, out: int) -> None:
        self._check(self._lib.sk_bam_flag_tlen_dev(self._h, flag, tid, mtid, tlen, n, max_frag, out),
                    "sk_bam_flag_tlen_dev")
