#!/usr/bin/env python3
"""How do N streams relate to a torch stream: same hardware queue (vgh_streams_overlap = 0) and, on different queues, do their
workgroups interleave (vgh_streams_interleave_permille ~1000) or does one kernel's dispatch block the other's (~500)?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from head_detector_amd import _lib  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
lib.vgh_streams_overlap.argtypes = [C.c_void_p, C.c_void_p]
lib.vgh_streams_interleave_permille.argtypes = [C.c_void_p, C.c_void_p]
lib.vgh_stream_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
torch.cuda.init()
main = torch.cuda.Stream()
x = torch.zeros(1, device="cuda")
cands = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    p = C.c_void_p()
    lib.vgh_stream_create(0, C.byref(p))
    cands.append(p)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("vs main:  overlap / interleave-permille")
for i, c in enumerate(cands):
    print(f"  cand {i}: {lib.vgh_streams_overlap(main.cuda_stream, c)} / {lib.vgh_streams_interleave_permille(main.cuda_stream, c)}")
print("pairwise interleave-permille among candidates (row = first kernel):")
for i, a in enumerate(cands):
    print("  ", " ".join(f"{lib.vgh_streams_interleave_permille(a, b):5d}" if i != j else "    -" for j, b in enumerate(cands)))
