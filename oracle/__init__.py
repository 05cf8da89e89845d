"""TEST INFRASTRUCTURE: CPU checkers for the HIP path (see nyx_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package."""
