"""Test suite: `-m "not gpu"` (oracle vs fixtures, host logic, C ABI, gloo sharding) and `-m gpu` (HIP parity)."""
