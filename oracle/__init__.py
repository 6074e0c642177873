"""TEST INFRASTRUCTURE: CPU restatements of the reference's algorithms (checkers only).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
