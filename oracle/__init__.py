"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/md_oracle.cpp header)."""
