"""TEST INFRASTRUCTURE — the CPU oracle for the MI355X backend.

Nothing under this package is part of the product. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
