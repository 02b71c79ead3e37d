"""oracle/ -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/sw_oracle.c (CPU restatement of the reference's
Stage-1 path) and helpers to run oracle/_ref/ref_driver (the reference's own
MASA-Core CPU path).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product never does.
"""
from .binding import *  # noqa: F401,F403
