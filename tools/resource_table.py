#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel the library ships, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage with the Makefile's flags).  No GPU needed.
    python tools/resource_table.py [--md]        -> table (markdown with --md) for DESIGN.md 5.1"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_by_phase import makefile_flags, CSRC

VARIANT = {0: "lean", 2: "lean + sun", 4: "lean + alpha", 6: "lean + alpha + sun"}


def remarks(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def pretty(name):
    m = re.search(r"path_pool_kernelILi(\d+)E", name)
    if m:
        f = int(m.group(1))
        what = VARIANT[f & 6].replace("lean", "materials" if f & 16 else "lean") + (", hbm-scene" if f & 8 else ", lds-scene") + (", deep stacks" if f & 32 else "") + (", statistics build" if f & 1 else "")
        return "path_pool<%d> (%s)" % (f, what)
    m = re.search(r"\d+(\w+_kernel)", name)
    return m.group(1) if m else name


def main():
    md = "--md" in sys.argv
    rows = []
    for src, extra in (("kernel_path_pool.hip", ()), ("kernel_path_pool.hip", ("-DDRT_POOL_EXT_TU",)), ("kernel_wave_queue.hip", ()),
                       ("render_kernels.hip", ()), ("kernel_bvh_build.hip", ())):
        for r in remarks(src, extra):
            if "VGPRs" in r and not (extra and "ILi" in r["name"] and int(re.search(r"ILi(\d+)E", r["name"]).group(1)) < 16):
                rows.append(r)
    cols = ["VGPRs", "VGPRs Spill", "TotalSGPRs", "SGPRs Spill", "ScratchSize", "Occupancy"]
    head = ["kernel"] + ["VGPR", "VGPR spills", "SGPR", "SGPR spills", "scratch B/lane", "waves/SIMD (registers)"]
    seen = set()
    lines = []
    for r in rows:
        key = pretty(r["name"])
        if "wave_queue" in key:
            key = re.sub(r"_ZN3drt12_GLOBAL__N_1\d+", "", r["name"])[:60]
        if key in seen:
            continue
        seen.add(key)
        lines.append([key] + [str(r.get(c, "")) for c in cols])
    if md:
        print("| " + " | ".join(head) + " |\n|" + "---|" * len(head))
        for l in lines:
            print("| " + " | ".join(l) + " |")
    else:
        for l in [head] + lines:
            print("%-62s %5s %12s %5s %12s %15s %s" % tuple(l))


if __name__ == "__main__":
    main()
