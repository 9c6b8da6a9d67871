"""CPU oracle for the fissure-segmentation hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline.  The product package
(``fissure_segmentation_amd``) never imports it and has no CPU fallback.

* ``oracle.c_api``   -- ctypes bindings of ``fsg_oracle.c`` (exact-arithmetic primitives).
* ``oracle.ref_cpu`` -- pure-PyTorch CPU restatement of the reference's dataflow (models, losses).
"""
