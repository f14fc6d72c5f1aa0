"""Builds csrc/*.hip into lib/libidh.so with hipcc for gfx950 (cross-compiles without a GPU).

Incremental: one object per source, re-compiled when the source or a header is newer.
Invoked by ``__graft_entry__.build()``; also usable as ``python implicit-depth_amd/build.py``.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "_obj")
LIB = os.path.join(LIBDIR, "libidh.so")

# -fvisibility=hidden: only what include/*.h declares (under `#pragma GCC visibility push(default)`) is exported
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# per-source extras.  conv_wino4: the SLP vectoriser packs the scalar transform FMAs into v_pk_fma_f32 behind v_mov shuffles (98 moves
# and 3 scratch reloads per two K stages; packed fp32 math is no faster than scalar beside fp32 MFMAs on gfx950)
# -amdgpu-prealloc-sgpr-spill-vgprs: the lanes that take spilled scalars are reserved before vector allocation; without it the late reservation
# fragments the file and the two staging quads held across the transform end up in scratch (tools/kernel_resources.py: 36 -> 8 B/lane, the 8 inside
# the epilogue)
EXTRA_FLAGS = {"conv_wino4.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"],
               "conv_wino4p.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"]}


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(verbose: bool = True, force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    jobs = []
    objs = []
    for s in sources:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[idh build]", " ".join(os.path.relpath(c, HERE) if os.path.isabs(c) and c.startswith(HERE) else c for c in cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
