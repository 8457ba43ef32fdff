"""CPU oracle for the tri-plane render + SR hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (real3dportrait_amd) never does.
"""
from .oracle import Oracle, build_oracle  # noqa: F401
