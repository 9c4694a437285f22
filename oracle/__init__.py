"""CPU oracle for the obman_train mesh-loss hot path.

TEST INFRASTRUCTURE ONLY.  A plain torch-CPU restatement of the reference's
algorithms (each function cites the reference file:line it follows).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it - as the checker, never as the thing measured or shipped.  The
product package ``obman_train_amd`` never imports this package and fails loudly
when its HIP library is missing.

Pinning: every function except the MANO layer is checked against golden vectors
produced by importing the reference itself in the dev container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  The MANO layer
lives in the un-vendored, un-pinned external package ``manopth`` (reference
call sites ``manobranch.py:6,92-105,170-182``) and needs licence-gated model
files: **MANO parity unpinned** - ``oracle.mano`` restates the published
MANO/SMPL formulation and is checked by analytic identities and fp64 finite
differences only.
"""
