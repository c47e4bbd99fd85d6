"""CPU oracle (test infrastructure).  See oracle/advchain_oracle.py."""
