"""ORACLE: CPU restatement of the reference hot path - test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; md_rdm_amd/ never does (tests/test_boundary.py enforces it)."""
