"""CPU oracle (TEST INFRASTRUCTURE ONLY — see oracle/mpm_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (taichi_mpm_amd) never does.
"""
