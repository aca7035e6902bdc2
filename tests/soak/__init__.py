"""Soak / fuzz runs and full-size checks against the oracle (run on the GPU box by hand:
`python tests/soak/fuzz_pooling.py 120`).  They live under tests/ because they use the oracle as
their checker; pytest does not collect them (no test_ prefix)."""
