"""ORACLE -- test infrastructure only.

CPU restatements of the reference's hot path used as the parity checker.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (pose-estimation_amd/) never does.
"""
