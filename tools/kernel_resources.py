"""Register / scratch / LDS / occupancy table of every kernel of csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the
flags of implicit-depth_amd/build.py).  Runs without a GPU.

  python tools/kernel_resources.py [source.hip ...] [--scratch-only] [--max-scratch kernel_substring=bytes ...]

`--max-scratch conv3x3_wino4_k=0` exits 1 when a kernel whose demangled name contains the substring uses more scratch than allowed
(tests/test_abi.py uses this for the kernels whose hot loops must stay spill-free).
"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "implicit-depth_amd"))
import build as idh_build  # noqa: E402

FIELDS = ("Name", "TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [o if o else n for o, n in zip(out, names)]
    except OSError:
        return names


def resources(src: str, extra_defs=()):
    cmd = [idh_build._hipcc(), *idh_build.FLAGS, *idh_build.EXTRA_FLAGS.get(os.path.basename(src), []), *extra_defs, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark: [^ ]+\s+(.*?): (.*?) \[-Rpass-analysis", line)
        if not m:
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(.*?): (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key.endswith("Function Name") or key == "Name":
            cur = {"Name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = demangle([r_["Name"] for r_ in rows])
    for r_, n in zip(rows, names):
        r_["Name"] = re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0]
    return rows


def main(argv):
    limits, srcs, scratch_only = {}, [], False
    it = iter(argv)
    for a in it:
        if a == "--max-scratch":
            continue
        if a == "--scratch-only":
            scratch_only = True
        elif "=" in a:
            k, v = a.split("=")
            limits[k] = int(v)
        else:
            srcs.append(a)
    if not srcs:
        srcs = sorted(glob.glob(os.path.join(idh_build.CSRC, "*.hip")))
    bad = 0
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
    for s in srcs:
        for r in resources(s):
            scratch = int(r.get("ScratchSize [bytes/lane]", 0))
            if scratch_only and scratch == 0:
                continue
            flag = ""
            for k, lim in limits.items():
                if k in r["Name"] and scratch > lim:
                    flag = f"  <-- exceeds {lim}"
                    bad += 1
            print(f"{r['Name'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {scratch:>8d} "
                  f"{r.get('Occupancy [waves/SIMD]', '?'):>4s} {r.get('LDS Size [bytes/block]', '?'):>7s}{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
