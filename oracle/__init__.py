"""CPU oracle for the infer_arvc hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package (the product path under streamvoiceanon_amd/ must fail loudly without the HIP
library and never falls back to this code).
"""
