"""CPU oracle for the NLML_HPE batched-inference hot path -- TEST INFRASTRUCTURE.

This package is a CPU restatement (numpy / torch-CPU / plain C) of the
reference's arithmetic for the path SURVEY.md section 8 scopes:

  oracle.feature_norm   helpers/FeatureExtractor.py:30-66     (IPD normalisation)
  oracle.encoder_heads  NLML_HPE_Model_Builder.py:26-126      (encoder + 3 heads)
  oracle.tucker         TD_Tester.py:25-58,162-199            (objective, Powell)
  oracle.metrics        NLML_HPE_Test.py:28-130               (MAE / MAEV / std)
  oracle.video_math     generatePose_on_video.py:73-124,215-224 (EMA, axes)
  oracle/csrc/oracle.c  plain-C restatement of the first three, fixed
                        summation order (built by oracle/Makefile)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The
product package ``nlml_hpe_amd`` never imports it and has no CPU fallback: it
raises if the HIP library is missing.

Pinning: the restatement is checked (tests/test_oracle_golden.py) against the
fixtures in tests/golden/, which were produced by importing the reference's own
Python from /root/reference in the build container
(tests/golden/make_golden.py is the generating script).  The reference itself
never travels to the GPU box.

Unpinned: pixels -> landmarks (MediaPipe FaceMesh, third party, not in the
reference tree, not installable here) -- the scope begins at raw landmarks.
"""
