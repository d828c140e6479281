"""CPU oracle for the HMC leapfrog path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``barcode_amd``) never does.  PARITY UNPINNED: see ``oracle/README.md``.
"""
