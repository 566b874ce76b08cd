"""CPU oracle for the PlatyMatch estimate_transform hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  platymatch_amd never does.  See pm_oracle.c / pm_oracle.py.
"""
from .pm_oracle import *  # noqa: F401,F403
