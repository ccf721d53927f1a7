"""CPU oracle for the RIME hot path -- test infrastructure only (see rime_oracle.py header)."""
