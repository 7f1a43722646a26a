"""CPU oracle for the CampX hot path.  TEST INFRASTRUCTURE - not part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  See campx_oracle.c for what it restates and how it is pinned.
"""
