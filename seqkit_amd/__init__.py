"""seqkit_amd — MI355X (gfx950) implementation of seqkit's per-read FASTQ hot path.

The product is the C-ABI shared library (include/seqkit_hip.h, built from seqkit_amd/csrc/)
and the C++ `fasta` / `sam` hosts above it.  This package is the thin Python view of that
C-ABI used by the tests and bench.py; it holds no arithmetic of its own and has no CPU
fallback: loading fails loudly when the HIP library is missing.
"""
from .capi import (Context, SeqkitHipError, library_path, load_library, SK_ASSIGN_AMBIGUOUS,  # noqa: F401
                   SK_ASSIGN_NONE, SK_DETAIL_FULL, SK_DETAIL_MATCHED, EXPORTED_SYMBOLS, BlockedLayout, blocked_layout)

__all__ = ["Context", "SeqkitHipError", "library_path", "load_library", "SK_ASSIGN_NONE",
           "SK_ASSIGN_AMBIGUOUS", "SK_DETAIL_FULL", "SK_DETAIL_MATCHED", "EXPORTED_SYMBOLS", "BlockedLayout", "blocked_layout"]
