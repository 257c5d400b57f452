"""CPU oracle for the Augmented Super-Resolution hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (torch-CPU / numpy, float32) of
the arithmetic the reference delegates to TensorFlow 2.7 / tensorflow-addons 0.15 (neither is
installable offline) plus the reference's own Python glue.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it -- and
only as the checker, never as the thing measured or shipped.  The product package
(``asr_amd``) never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and
TF/TFA cannot be imported here (ModuleNotFoundError, no network), so this restatement is
pinned only by known-answer tests derived from the documented semantics of the pinned TF/TFA
versions -- not by outputs of the reference itself.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
