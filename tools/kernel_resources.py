#!/usr/bin/env python
"""Per-kernel register / spill / LDS / code-size table of the gfx950 code objects (no GPU needed).

  python tools/kernel_resources.py [pps_k3.hip ...] [--filter REGEX] [--flags "..."]

Compiles every given .hip file of pop_up_slam_amd/csrc device-only (the Makefile's flags), unbundles the gfx950 ELF and
reads the AMDGPU metadata notes.  waves/SIMD = floor(512 / (vgpr_count rounded up to 8)) -- vgpr_count is the unified total, AGPRs included --, capped at 8 (MI355X_MICROARCH.md).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pop_up_slam_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
SOLVER = {"pps_k1.hip", "pps_k2.hip", "pps_k3.hip", "pps_k4.hip", "pps_dense.hip"}


def resources(src, extra):
    with tempfile.TemporaryDirectory() as td:
        bundle, elf = os.path.join(td, "k.bundle"), os.path.join(td, "k.elf")
        fp = "-ffp-contract=fast-honor-pragmas" if os.path.basename(src) in SOLVER else "-ffp-contract=off"   # (the Makefile's SOLVER_FP)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", fp, "--cuda-device-only", "-c", src, "-o", bundle] + extra
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + bundle, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--output=" + elf, "--unbundle"])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", elf], text=True)
        syms = subprocess.check_output([LLVM + "/llvm-readelf", "-s", "--wide", elf], text=True)
    size = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(name|vgpr_count|agpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "agpr_count":                     # first key of a kernel record
            cur = {}
            out.append(cur)
        if k == "name" and not line.startswith("    .name"):   # (argument names sit deeper)
            continue
        cur[k] = v
    for r in out:
        r["code_bytes"] = size.get(r["name"], 0)
    return out


def demangle(n):
    try:
        return subprocess.check_output(["c++filt", n], text=True).strip()
    except Exception:
        return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*", default=["pps_k1.hip", "pps_k2.hip", "pps_k3.hip", "pps_k4.hip"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    print(f"{'kernel':<78} {'vgpr':>5} {'agpr':>5} {'w/SIMD':>6} {'sspill':>6} {'vspill':>6} {'scratch':>7} {'lds':>6} {'code':>7}")
    for f in a.files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        for r in sorted(resources(path, a.flags.split()), key=lambda r: r["name"]):
            name = re.sub(r"\(.*", "", demangle(r["name"])).replace("pps::", "").replace("void ", "")
            if a.filter and not re.search(a.filter, name):
                continue
            v, ag = int(r["vgpr_count"]), int(r["agpr_count"])
            tot = (v + 7) // 8 * 8                   # .vgpr_count is the unified total (AGPRs included)
            print(f"{name:<78} {v:>5} {ag:>5} {min(8, 512 // max(8, tot)):>6} {r['sgpr_spill_count']:>6} {r['vgpr_spill_count']:>6} "
                  f"{r['private_segment_fixed_size']:>7} {r['group_segment_fixed_size']:>6} {r['code_bytes']:>7}")


if __name__ == "__main__":
    sys.exit(main())
