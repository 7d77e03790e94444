"""CPU oracle (test infrastructure, not the product).  See oracle/bdrt_oracle.h."""
