"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference VGGHeads forward path.

Nothing in ``head_detector_amd`` (the product) may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / reported baseline -- never as the thing measured or shipped.

Pinning status (see DESIGN.md "Oracle"):
  * flame_oracle   -- pinned against the *imported* reference glue
                      (head_detector/flame.py, utils.py, head_info.py run in the build
                      container around this package's ``lbs``; vectors in tests/golden/)
                      and against the reference's only known-answer fixture
                      (yolo_head_training/tests/1.json) for parameter layout + rigid stage.
                      The inner ``smplx.lbs.lbs`` arithmetic is third-party (smplx==0.1.26,
                      requirements.txt:7), absent from /root/reference and from this image:
                      its published algorithm is restated here => that part is "parity unpinned".
  * nms_oracle     -- torchvision ~=0.15.2 ``ops.boxes.nms`` (requirements.txt:2) is absent:
                      published CPU algorithm restated; the surrounding glue
                      (head_detector/utils.py:159-194) is pinned by import.  "parity unpinned"
                      for the inner greedy loop.
  * raster_oracle  -- Sim3DR rasteriser / PNCC composition / refined_head_bbox: PINNED.  The reference's only native
                      code (head_detector/Sim3DR/lib/rasterize_kernel.cpp) compiles from its own two files, so
                      ``oracle/build_ref.py`` builds it where it lies into ``oracle/_ref/libsim3dr_ref.so`` and the
                      restatement is bit-identical to it; PNCCProcessor / refined_head_bbox vectors come from the
                      reference's own Python run around that library (tests/golden/make_golden.py (f)).
  * letterbox_oracle -- cv2.resize(INTER_LANCZOS4) + copyMakeBorder (opencv-python, requirements.txt, version unpinned) are
                      absent: OpenCV's published 8-bit fixed-point algorithm restated.  "parity unpinned".
  * net_oracle     -- super_gradients>=3.7 block definitions are absent and no weights are
                      reachable: architecture restated from the arch YAMLs + SG semantics.
                      "parity unpinned" (the reference's own test is ``assert True``,
                      yolo_head_training/tests/test_models.py:29-36).
"""
