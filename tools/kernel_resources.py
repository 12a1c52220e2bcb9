#!/usr/bin/env python
"""Per-kernel register / spill / LDS / code-size table of the gfx950 code objects (no GPU needed).

  python tools/kernel_resources.py [pps_k3.hip ...] [--filter REGEX] [--flags "..."]
  python tools/kernel_resources.py --built            # the code objects inside csrc/build/*.hip.o -- what libpps.so ships; no compile
  python tools/kernel_resources.py --update-design    # rewrites the register table of DESIGN.md section 5 from the built objects

Compiles every given .hip file of pop_up_slam_amd/csrc device-only (the Makefile's flags) -- or, with --built, takes the .hip_fatbin
section of the object the Makefile built --, unbundles the gfx950 ELF into a temporary directory and reads the AMDGPU metadata notes.
tests/test_kernel_table.py compares DESIGN.md's table with the --built figures (8 VGPRs of slack, same waves / SIMD).  waves/SIMD = floor(512 / (vgpr_count rounded up to 8)) -- vgpr_count is the unified total, AGPRs included --, capped at 8 (MI355X_MICROARCH.md).
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
SOLVER = {"pps_k1.hip", "pps_k2.hip", "pps_k3.hip", "pps_dense.hip"}      # (built with contraction; pps_k1_lanes.hip / pps_k4.hip without)


def resources(src, extra, built=False):
    with tempfile.TemporaryDirectory() as td:
        bundle, elf = os.path.join(td, "k.bundle"), os.path.join(td, "k.elf")
        if built:
            obj = os.path.join(CSRC, "build", os.path.basename(src) + ".o")
            subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, bundle])
        else:
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


def short_name(mangled):
    return re.sub(r"\(.*", "", demangle(mangled)).replace("pps::", "").replace("void ", "")


# the kernels of DESIGN.md section 5's register table, in its order: (name as the tool prints it, what the row says in addition)
DESIGN_KERNELS = [
    ("k_trial_lin", "K4 + K1: both trials and the predicted lane-form sweep, C2's launch per LM pair"), ("k_linearize_lanes", "K1, lane form on its own"),
    ("k_linearize_obs_numeric", "K1, thread form: plane observations"),
    ("kb_linearize<0, 0, true>", ""), ("kb_linearize<0, 1, false>", "odometry / priors, numeric"),
    ("k_hblocks2", "K2"), ("k_hfinish", ""), ("kb_hblocks_tc", "K2 of large batches: class bodies"), ("kb_hblocks_tg", "... the generic body"),
    ("k_band_factor_all", "the whole tree in one launch: C2, C5"), ("k_band_solve_all", "... and its back-substitution"),
    ("k_band_factor_pre", "one band per launch, pre-assembling walk: C3"), ("k_band_root<true, true>", "top group: pre-assembling walk + data-flow back-substitution in one launch"), ("k_band_root<false, false>", "top group, plain walk / barrier form"),
    ("k_band_solve_flow", "data-flow back-substitution"), ("k_band_solve", "barrier form"),
    ("k_band_factor<true>", "plain walk"), ("k_band_factor<false>", "general: traces"), ("k_band_factor_r5", "fronts of 65-80 rows"),
    ("kb_band_factor<true>", ""), ("kb_band_factor_pre", ""), ("kb_level_factor2", ""), ("kb_level_factor3", ""), ("kb_level_factor4", ""),
    ("kb_level_solve", ""), ("k_trial_dual", "K4"), ("k_chi2", ""),
]
TABLE_BEGIN, TABLE_END = "<!-- kernel-table:begin (tools/kernel_resources.py --update-design) -->", "<!-- kernel-table:end -->"


def built_table():
    """{name: dict} of every kernel of the solver objects the Makefile built"""
    out = {}
    for f in ("pps_k1.hip", "pps_k1_lanes.hip", "pps_k2.hip", "pps_k3.hip", "pps_k4.hip"):
        for r in resources(os.path.join(CSRC, f), [], built=True):
            v = int(r["vgpr_count"])
            out[short_name(r["name"])] = {"vgpr": v, "agpr": int(r["agpr_count"]), "waves": min(8, 512 // max(8, (v + 7) // 8 * 8)),
                                          "sspill": int(r["sgpr_spill_count"]), "vspill": int(r["vgpr_spill_count"]),
                                          "lds": int(r["group_segment_fixed_size"]), "code": int(r["code_bytes"])}
    return out


def design_rows(tab):
    rows = ["| kernel | VGPR | AGPR | waves / SIMD | SGPR spills | VGPR spills | code bytes |", "|---|---|---|---|---|---|---|"]
    for name, note in DESIGN_KERNELS:
        if name not in tab:
            continue
        r = tab[name]
        rows.append(f"| `{name}`{' (' + note + ')' if note else ''} | {r['vgpr']} | {r['agpr']} | {r['waves']} | {r['sspill']} | {r['vspill']} | {r['code']} |")
    return rows


def parse_design_table(text):
    """rows of the table between the markers -> {kernel name: (vgpr, agpr, waves, sspill, vspill, code)}"""
    body = text[text.index(TABLE_BEGIN) + len(TABLE_BEGIN):text.index(TABLE_END)]
    out = {}
    for line in body.splitlines():
        m = re.match(r"\| `([^`]+)`[^|]*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|", line)
        if m:
            out[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    return out


def update_design():
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    i, j = text.index(TABLE_BEGIN) + len(TABLE_BEGIN), text.index(TABLE_END)
    text = text[:i] + "\n" + "\n".join(design_rows(built_table())) + "\n" + text[j:]
    open(path, "w").write(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*", default=["pps_k1.hip", "pps_k1_lanes.hip", "pps_k2.hip", "pps_k3.hip", "pps_k4.hip"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--flags", default="")
    ap.add_argument("--built", action="store_true", help="read the objects under csrc/build instead of compiling")
    ap.add_argument("--update-design", action="store_true")
    a = ap.parse_args()
    if a.update_design:
        return update_design()
    print(f"{'kernel':<78} {'vgpr':>5} {'agpr':>5} {'w/SIMD':>6} {'sspill':>6} {'vspill':>6} {'scratch':>7} {'lds':>6} {'code':>7}")
    for f in a.files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        for r in sorted(resources(path, a.flags.split(), built=a.built), key=lambda r: r["name"]):
            name = re.sub(r"\(.*", "", demangle(r["name"])).replace("pps::", "").replace("void ", "")
            if a.filter and not re.search(a.filter, name):
                continue
            v, ag = int(r["vgpr_count"]), int(r["agpr_count"])
            tot = (v + 7) // 8 * 8                   # .vgpr_count is the unified total (AGPRs included)
            print(f"{name:<78} {v:>5} {ag:>5} {min(8, 512 // max(8, tot)):>6} {r['sgpr_spill_count']:>6} {r['vgpr_spill_count']:>6} "
                  f"{r['private_segment_fixed_size']:>7} {r['group_segment_fixed_size']:>6} {r['code_bytes']:>7}")


if __name__ == "__main__":
    sys.exit(main())
