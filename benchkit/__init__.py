"""Benchmark plumbing shared by bench.py and tools/ (workload builders of BASELINE configs 3-5, timed regions, CPU-baseline legs).

Not part of the product package: the CPU-baseline legs time `oracle/` (the checker), which nothing under rabe_amd/ may import."""
