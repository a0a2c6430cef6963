"""CPU oracle (test infrastructure only; see stnerf_oracle.py)."""
