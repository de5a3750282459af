"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see mmx_oracle.c).  Parity unpinned: see DESIGN.md."""
