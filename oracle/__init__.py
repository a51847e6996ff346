"""CPU oracle for the facet-graph-convolution hot path: TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
