"""CPU oracle for the MedicalSeg VNet hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product
package ``medicalseg_amd`` never imports this package and fails loudly when
its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * model / loss / optimizer restatement (``vnet_numpy.py``, ``c/``):
    PARITY UNPINNED BY THE REFERENCE -- PaddlePaddle is not vendored in
    /root/reference, cannot be installed here, and the reference ships no
    tests or golden vectors for this path.  The restatement is cross-checked
    against torch-CPU functional ops (the reference itself was ported from a
    torch implementation, models/vnet.py:1-3) and against the loss
    known-answer vector of SURVEY.md Appendix C.
  * preprocessing restatement (``preprocess_numpy.py``): PINNED against
    outputs of the reference's own ``tools/preprocess_utils`` code imported in
    the build container (tests/golden/make_preprocess_golden.py).
"""
