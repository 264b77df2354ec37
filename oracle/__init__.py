"""CPU oracle -- test infrastructure only (see oracle/ref_numpy.py header for the import rule)."""
