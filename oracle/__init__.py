"""CPU oracle (test infrastructure only -- never imported by loudgain_amd)."""
