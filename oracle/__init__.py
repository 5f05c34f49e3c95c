"""Parity oracle: TEST INFRASTRUCTURE ONLY (see oracle/fe_oracle.c header)."""
