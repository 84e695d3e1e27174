"""CPU oracle for the optical-flow hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy/Pillow restatement of the reference's Numba/SciPy CPU algorithms (and of
its OpenCL kernels), call for call, used only as the checker by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under
``opticalflow-ri_amd/`` imports this package; the product path is HIP only.

Parity status: the Horn-Schunck, Gaussian pre-filter, pyramid driver and Liu-Shen
restatements are pinned against fixtures produced by running the reference itself
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).  The Farneback and dense
Lucas-Kanade restatements follow the reference's OpenCL source text; the reference
cannot execute those kernels in the build environment (no OpenCL device), so for them
only the host constant tables are pinned and the kernels are "parity unpinned".
"""
