"""In-tree build of libvgh.so (hipcc, gfx950 only). `python -m head_detector_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvgh.so")
SOURCES = ["conv_igemm.hip", "conv_patch.hip", "conv_rings.hip", "conv_pp.hip", "ds_b2b.hip", "conv_split.hip", "conv_f32.hip", "stem_pool.hip", "postproc.hip", "flame.hip", "net.hip", "detect.hip", "raster.hip", "letterbox.hip", "ctx.hip", "streams.hip"]
EXPERIMENT_SOURCES = ["stem_ds.hip"]  # measured losers kept for tools/: part of libvgh_exp.so (-DVGH_EXPERIMENTS) only
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vgh.h")]
    return any(os.path.getmtime(d) > t for d in deps)


LIB_EXP = os.path.join(HERE, "libvgh_exp.so")  # -DVGH_EXPERIMENTS build (work-skipping switches, env-var knobs): tools/ only


def build_lib(force: bool = False, verbose: bool = True, experiments: bool = False, variant_defines=None) -> str:
    if variant_defines:  # A/B builds of the PRODUCT code with one compile-time knob changed (no experiment switches): libvgh_var.so
        return _build(os.path.join(HERE, "libvgh_var.so"), [f"-D{d}" for d in variant_defines], "build_var", verbose)
    if experiments:
        return _build(LIB_EXP, ["-DVGH_EXPERIMENTS"], "build_exp", verbose)
    if not force and not needs_build():
        return LIB
    return _build(LIB, [], "build", verbose)


def _build(LIB: str, extra, objdir: str, verbose: bool) -> str:
    hipcc = _hipcc()
    objs, procs = [], []
    t0 = time.time()
    os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
    for src in SOURCES + (EXPERIMENT_SOURCES if "-DVGH_EXPERIMENTS" in extra else []):
        obj = os.path.join(HERE, objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[vgh build] {src} FAILED\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(f"[vgh build] {src}:\n{out}\n")
    if failed:
        raise RuntimeError("libvgh.so: compilation failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"libvgh.so: link failed\n{r.stdout}")
    import ctypes

    try:  # catches undefined symbols (e.g. a kernel whose host stub was silently dropped) at build time, not on the GPU box
        ctypes.CDLL(LIB)
    except OSError as e:
        os.remove(LIB)
        raise RuntimeError(f"libvgh.so: built but does not load: {e}")
    if verbose:
        sys.stderr.write(f"[vgh build] built {LIB} in {time.time() - t0:.1f}s\n")
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, experiments="--experiments" in sys.argv, variant_defines=[a[2:] for a in sys.argv[1:] if a.startswith("-D")])
