"""CPU oracle for the rabe hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything
from this package.  The product (rabe_amd/) never does; it fails loudly without its HIP
extension.
"""
