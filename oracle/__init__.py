"""CPU oracle for the marlhip hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package.  Nothing under ``codebase_amd/``
imports it: the product path runs on the HIP library or fails loudly.

Parity status:
  * learner half (ReplayBuffer / QNetwork / eps schedule): PINNED - checked
    against the reference's own classes imported from /root/reference by
    ``oracle/make_golden.py``; vectors frozen under ``tests/golden/``.
  * env half (lbforaging step/reset): PARITY UNPINNED - the third-party
    ``lbforaging`` package is not vendored in the reference, not installed and
    not fetchable; ``oracle/lbf.py`` restates its published algorithm.
"""
