"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's EmbeddingBag hot path (see embbag_oracle.c
and embbag_oracle.py headers for the reference file:line each routine follows).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; param_amd/ never does.
"""
