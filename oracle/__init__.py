"""Test infrastructure only: CPU oracles for the TSDF hot path (see oracle/tsdf_oracle.c header)."""
