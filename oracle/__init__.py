"""oracle/ -- CPU restatement of the reference algorithms. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker or as the timed CPU baseline.  The
product package (efficientlo-net_amd/) never imports it.
"""
