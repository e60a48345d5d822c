"""TEST INFRASTRUCTURE.  Builds oracle/_ref/libsim3dr_ref.so = the reference's own Sim3DR rasteriser
(/root/reference/head_detector/Sim3DR/lib/rasterize_kernel.cpp, compiled from where it lies with the flags of the
reference's setup.py: -std=c++11, head_detector/Sim3DR/setup.py:20) + oracle/ref_shim.cpp (C-linkage forwarders).

The reference's Cython binding (rasterize.pyx) is NOT built: it needs Cython-generated code; the two C++ functions it
forwards to are called directly instead.  Outputs go to oracle/_ref/ only (git-ignored, travels to the GPU box with
gpurun like our own .so files).  On a machine without /root/reference the prebuilt library is used as is."""
import ctypes
import os
import subprocess
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/head_detector/Sim3DR/lib/rasterize_kernel.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libsim3dr_ref.so")


def build(verbose: bool = False) -> Optional[str]:
    """Returns the library path, or None when neither the reference sources nor a prebuilt library exist."""
    if os.path.exists(REF_SRC):
        shim = os.path.join(HERE, "ref_shim.cpp")
        stale = not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(REF_SRC), os.path.getmtime(shim))
        if stale:
            os.makedirs(OUT_DIR, exist_ok=True)
            # -ffp-contract=off: plain IEEE float ops, what an x86-64 baseline build of the reference executes (no FMA without -mfma)
            cmd = ["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", REF_SRC, shim, "-o", LIB]
            if verbose:
                print("[oracle/_ref]", " ".join(cmd))
            subprocess.check_call(cmd)
    return LIB if os.path.exists(LIB) else None


def load() -> Optional[ctypes.CDLL]:
    path = build()
    if path is None:
        return None
    lib = ctypes.CDLL(path)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.ref_rasterize.argtypes = [P, P, P, P, P, I, I, I, I, F, I]
    lib.ref_rasterize.restype = None
    lib.ref_rasterize_triangles.argtypes = [P, P, P, P, P, I, I, I]
    lib.ref_rasterize_triangles.restype = None
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
