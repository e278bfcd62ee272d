"""CPU parity oracle for the `bce -c` path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (bce_amd/) never does.  See oracle/bce_oracle.c for the restatement itself.
"""
from .oracle import *  # noqa: F401,F403
