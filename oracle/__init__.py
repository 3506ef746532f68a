"""CPU oracle (test infrastructure only; parity unpinned — see oracle/seqkit_oracle.h)."""
