"""TEST INFRASTRUCTURE: CPU oracle (checker) for the HIP engine.  Not product code."""
