"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch-CPU fp32 / fp64, functional style) of the MIMRL two-stage
training step.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product package ``mimrl_amd`` never imports anything from here and fails loudly
when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference
(/root/reference, dev container only) and captures outputs on seeded inputs; the
restatement is checked against those fixtures in ``tests/test_oracle_vs_golden.py``.
The reference ships no golden vectors of its own.
"""
