"""CPU oracle for the multigrid V-cycle path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the timed CPU baseline.  Nothing under
``multigridcmt_amd/`` imports it; the product path raises when the HIP library is missing.

Contents
--------
``sparse_ref``   NumPy/SciPy restatement of the reference's algorithm on general ``scipy.sparse``
                 operators (same semantics as MGCMTSolver.py / MGCMTStencilMaker.py /
                 MGCMTProcessor.py, O(N) per sweep instead of the reference's O(N^2) setup).
``structured``   matrix-free restatement on tensor-product tridiagonal factors (NumPy front end
                 of ``mgcmt_oracle.c``; the C file is built by ``oracle/Makefile``).
``ref_loader``   container-only loader that runs the Python-2 reference itself (used by
                 ``gen_golden.py`` to write ``tests/golden/*.npz``).

Parity status: PINNED — ``sparse_ref`` is checked against the reference's own known-answer
values (UnitTests/*.py) and against golden vectors produced by running the reference in the
build container (``tests/golden/``, generator ``oracle/gen_golden.py``).
"""
