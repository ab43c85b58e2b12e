#!/usr/bin/env python3
"""Static instruction census of the shipped path_pool kernel, by phase and by reference operation.

Compiles dustraytracer_amd/csrc/kernel_path_pool.hip exactly as the Makefile does (plus -gline-tables-only, which does
not change code generation), disassembles the requested variant, asks llvm-symbolizer for the inline stack of every
instruction and files the instruction under
  * the PHASE whose source region the outermost frame in kernel_path_pool.hip lies in (regions are found from the
    `// ============ X:` banners, the claim banner and the push_group call), and
  * the OPERATION = the innermost device_math.hpp / device_access.hpp function it was inlined from
    (tri_intersect_flat = Intersection.cu:4-36, slab_entry_or_inf = Bounds.cu:18-41, random_unit_sphere_try =
    Random.cu:50-58, ...), "own" for code written in the kernel body itself.

VALU issue slots follow bench.py's pricing: 1 per VALU instruction, 2 for v_fma / v_mad, 4 for the transcendental unit.
Output: JSON (--json) and a table.  tools/roofline_from_profiles.py uses the per-operation slots for `algorithmic_frac`.

No GPU needed.  Usage: python tools/isa_by_phase.py [--flags 0] [--json out.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dustraytracer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(CSRC, "kernel_path_pool.hip")


def makefile_flags():
    """COMMON + DEVFLAGS of the Makefile, so that the census is of the shipped code."""
    text = open(os.path.join(CSRC, "Makefile")).read()
    common = [f for f in re.search(r"^COMMON\s*:=\s*(.*)$", text, re.M).group(1).split() if not f.startswith("$(")]
    dev = re.search(r"^DEVFLAGS\s*:=\s*(.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    return common + dev


def phase_regions():
    """[(first_line, last_line, phase)] from the banners of the kernel source."""
    lines = open(SRC).read().split("\n")
    marks = []
    for i, l in enumerate(lines, 1):
        m = re.search(r"// =+ ([A-Z][0-9A-Za-z]*)\b.*=+\s*$", l)
        if m:
            marks.append((i, m.group(1)))
        elif "// ============ the N loop" in l:
            marks.append((i, "NLOOP"))
        elif "---------------- choose a queue" in l:
            marks.append((i, "claim"))
        elif re.search(r"^\s+if \(!HBM\) __builtin_amdgcn_s_setprio\(1\);\s*$", l) and marks and marks[-1][1] not in ("claim", "prologue"):
            marks.append((i, "push"))
        elif "---- prologue" in l:
            marks.append((i, "prologue"))
        elif re.search(r"^\s+for \(;;\) \{\s*$", l) and marks and marks[-1][1] == "prologue":
            marks.append((i, "claim"))
        elif "fp.span[1]" in l:
            marks.append((i, "epilogue"))
    regions = []
    for k, (start, name) in enumerate(marks):
        end = marks[k + 1][0] - 1 if k + 1 < len(marks) else len(lines)
        regions.append((start, end, name))
    return regions


def classify(mnemonic):
    if mnemonic.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"                      # SGPR spill / broadcast traffic: VALU slots, no arithmetic
    if mnemonic.startswith("v_"):
        return "valu"
    if mnemonic.startswith("ds_"):
        return "lds"
    if mnemonic.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mnemonic.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if mnemonic.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_barrier")):
        return "wait"
    if mnemonic.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_getpc")):
        return "branch"
    if mnemonic.startswith("s_"):
        return "salu"
    return "other"


def slots(mnemonic):
    if not mnemonic.startswith("v_"):
        return 0
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", mnemonic):
        return 4
    if re.match(r"v_(fma|mad|fmac|mac|pk_fma)_", mnemonic):
        return 2
    return 1


def census(flags, keep=None):
    tmp = tempfile.mkdtemp(prefix="isa_by_phase_")
    elf = os.path.join(tmp, "pp.elf")
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + ["-gline-tables-only", "--cuda-device-only", "--no-gpu-bundle-output",
                                                         "-c", SRC, "-o", elf]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
    want = "path_pool_kernelILi%dE" % flags
    insts = []                                  # (address, mnemonic)
    inside = False
    for line in dis.split("\n"):
        m = re.match(r"^([0-9a-f]{16}) <(.*)>:", line)
        if m:
            inside = want in m.group(2)
            continue
        if not inside:
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]{12}):", line)
        if m:
            insts.append((int(m.group(2), 16), m.group(1)))
    if not insts:
        raise SystemExit("kernel variant %d not found" % flags)
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + elf, "--inlines", "--functions=short"],
                         input="\n".join(hex(a) for a, _ in insts) + "\n", capture_output=True, text=True, check=True).stdout
    stacks = [blk.strip().split("\n") for blk in sym.strip().split("\n\n")]
    assert len(stacks) == len(insts), (len(stacks), len(insts))
    regions = phase_regions()

    def phase_of(line):
        for a, b, name in regions:
            if a <= line <= b:
                return name
        return "setup"

    table = collections.defaultdict(lambda: collections.Counter())
    sites = collections.defaultdict(set)
    for (addr, mn), st in zip(insts, stacks):
        frames = [(st[i], st[i + 1]) for i in range(0, len(st) - 1, 2)]      # (function, file:line:col), innermost first
        outer_line = 0
        for fn, loc in frames:                                              # outermost frame inside the kernel source
            f, l = loc.rsplit(":", 2)[0:2]
            if f.endswith("kernel_path_pool.hip"):
                outer_line = int(l)
        op = "own"
        for fn, loc in frames:
            f = loc.rsplit(":", 2)[0]
            if f.endswith(("device_math.hpp", "device_access.hpp")):
                op = fn
        # the operation that CALLED a helper counts (exact_rcp inside tri_intersect_flat is the triangle test's)
        for fn, loc in frames:
            f = loc.rsplit(":", 2)[0]
            if f.endswith(("device_math.hpp", "device_access.hpp")) and fn in (
                    "tri_intersect_flat", "slab_entry_or_inf", "slab_intersect", "random_unit_sphere_try", "random_unit_vec3",
                    "camera_get_ray", "closest_hit_frame", "sky_model", "uncharted2_filmic", "gamma_correction", "tex_get_pixel",
                    "any_hit", "make_ray", "accumulate_and_resolve", "interp_uv"):
                op = fn
        ph = phase_of(outer_line)
        c = table[(ph, op)]
        c[classify(mn)] += 1
        c["slots"] += slots(mn)
        # the call site of `op` (the location in the frame just outside it): distinct call sites = inlined copies
        for i, (fn, loc) in enumerate(frames):
            if fn == op and i + 1 < len(frames):
                sites[(ph, op)].add(frames[i + 1][1])
                break
    for key, c in table.items():
        c["copies"] = max(1, len(sites.get(key, ())))
    return table, len(insts)


# ---- per-operation cost: tools/probes/isa_probes.hip, one kernel per reference operation, compiled with the Makefile's flags ----
COLD_LINES = None


def cold_lines():
    """device_math.hpp lines whose code runs only outside 2^-100..2^100 (the plain operator behind exact_rcp / exact_sqrt)."""
    global COLD_LINES
    if COLD_LINES is None:
        COLD_LINES = set()
        for i, l in enumerate(open(os.path.join(CSRC, "device_math.hpp")).read().split("\n"), 1):
            if re.search(r"__builtin_expect\(.*\) q = 1\.0f / x;|return sqrtf\(x\);", l):
                COLD_LINES.add(i)
    return COLD_LINES


def probe_slots():
    """{operation: VALU issue slots of one call} from the probe kernels (cold fallback paths excluded, empty probe subtracted)."""
    tmp = tempfile.mkdtemp(prefix="isa_probes_")
    elf = os.path.join(tmp, "probes.elf")
    src = os.path.join(ROOT, "tools", "probes", "isa_probes.hip")
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + ["-gline-tables-only", "--cuda-device-only", "--no-gpu-bundle-output", "-c", src, "-o", elf]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
    per = collections.OrderedDict()
    cur = None
    for line in dis.split("\n"):
        m = re.match(r"^([0-9a-f]{16}) <(.*)>:", line)
        if m:
            k = re.search(r"probe_(\w+?)(?:PK|P[fK]|Pf)", m.group(2))
            cur = k.group(1) if k else None
            if cur:
                per[cur] = []
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-F]{12}):", line)
        if m and cur:
            per[cur].append((int(m.group(2), 16), m.group(1)))
    out = {}
    for name, insts in per.items():
        sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + elf, "--inlines", "--functions=short"],
                             input="\n".join(hex(a) for a, _ in insts) + "\n", capture_output=True, text=True, check=True).stdout
        stacks = [blk.strip().split("\n") for blk in sym.strip().split("\n\n")]
        total = cold = 0
        for (addr, mn), st in zip(insts, stacks):
            loc = st[1] if len(st) > 1 else ""
            f, l = (loc.rsplit(":", 2) + ["0", "0"])[0:2]
            is_cold = f.endswith("device_math.hpp") and int(l) in cold_lines()
            if is_cold:
                cold += slots(mn)
            else:
                total += slots(mn)
        out[name] = (total, cold)
    base = out.pop("empty")[0]
    return {k: {"slots": v[0] - base, "cold_fallback_slots": v[1]} for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=0, help="kernel variant: 1 stats, 2 sunlight, 4 alpha, 8 hbm-scene")
    ap.add_argument("--json", default=None)
    ap.add_argument("--slots-json", default=None, help="write the per-operation VALU issue slots of every production variant (0, 2, 4, 6, 8, 10, 12, 14)")
    args = ap.parse_args()
    if args.slots_json:
        probes = probe_slots()
        t = {"_doc": ("VALU issue slots (1 = 2 SIMD cycles; fma / mad 2, rcp / sqrt 4) of ONE call of each reference operation: the kernels of "
                      "tools/probes/isa_probes.hip (the same device_math.hpp / device_access.hpp functions the tracing kernel inlines), compiled with the "
                      "Makefile's flags, disassembled and counted by tools/isa_by_phase.py; the empty probe is subtracted, the out-of-range fallbacks of "
                      "exact_rcp / exact_sqrt are listed apart.  bench.py: algorithmic_frac = sum(work counter x slots) / 64 lanes / kernel time / peak. "
                      "in_kernel: the same helpers as inlined in path_pool_kernel<0> (static slots of all their inlined copies, fallbacks included) for comparison."),
             "per_call": probes}
        table, n = census(0)
        t["in_kernel"] = {"%s/%s" % (ph, op): {"slots": c["slots"], "source_call_sites": c["copies"]} for (ph, op), c in sorted(table.items())
                          if op in ("tri_intersect_flat", "slab_entry_or_inf", "random_unit_sphere_try", "closest_hit_frame", "make_ray", "camera_get_ray")}
        with open(args.slots_json, "w") as f:
            json.dump(t, f, indent=1)
        for k, v in probes.items():
            print("%-16s %s" % (k, v))
        return
    table, n = census(args.flags)
    cols = ["valu", "lane", "slots", "salu", "lds", "vmem", "smem", "wait", "branch"]
    print("path_pool_kernel<%d>: %d instructions (static)" % (args.flags, n))
    print("%-9s %-26s " % ("phase", "operation") + " ".join("%6s" % c for c in cols))
    by_phase = collections.defaultdict(collections.Counter)
    for (ph, op), c in sorted(table.items()):
        print("%-9s %-26s " % (ph, op) + " ".join("%6d" % c[k] for k in cols))
        by_phase[ph].update(c)
    print()
    for ph, c in sorted(by_phase.items()):
        print("%-9s %-26s " % (ph, "(all)") + " ".join("%6d" % c[k] for k in cols))
    if args.json:
        out = {"flags": args.flags, "instructions": n,
               "by_phase_op": [{"phase": ph, "op": op, **{k: c[k] for k in cols}} for (ph, op), c in sorted(table.items())]}
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
