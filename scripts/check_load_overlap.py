"""Static check of the generated gfx950 code: within one group of vector-memory loads that are in flight together (no
s_waitcnt vmcnt between them), no load may write registers that an earlier load of the group writes.  The register
allocator produces this when part of a wide load is dead (it hands the dead registers to the next load); the hardware
then cannot overlap the two loads, which cost the out-of-cache configurations 12 % (profiles/r02s_c4_regression.txt).

The same pass checks the instruction budget of the packet walk's node step (packet_step_budget).

usage: check_load_overlap.py [extra hipcc flags]      exit status 1 if an overlap is found or a packet node step is over budget"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def makefile_flags():
    text = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    kflags = re.search(r"^KFLAGS\s*\??=\s*(.*)$", text, re.M)          # kernels.hip only (scheduling strategy)
    return flags + (kflags.group(1).split() if kflags else [])


def find_overlaps(asm_text):
    """[(kernel, first load, second load)] for every pair of loads in one in-flight group with intersecting destinations"""
    out = []
    kernel = None
    group = []          # (text, set of dest vgprs)
    for line in asm_text.split("\n"):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, group = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".LBB") or s.startswith("; %bb."): group = []
            continue
        op = s.split()[0]
        if op.startswith("s_waitcnt") and "vmcnt" in s or op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            group = []
            continue
        if op.startswith(("global_load", "flat_load", "buffer_load")):      # (scratch reloads are the register allocator's own)
            d = re.match(r"^\S+\s+v\[(\d+):(\d+)\]|^\S+\s+v(\d+)", s)
            if not d: continue
            regs = set(range(int(d.group(1)), int(d.group(2)) + 1)) if d.group(1) else {int(d.group(3))}
            for prev, pregs in group:
                if regs & pregs: out.append((kernel, prev, s))
            group.append((s, regs))
    return out


def packet_step_budget(asm_text, max_other=52, max_valu=62):
    """[(kernel, block, n_valu, n_other)] for node-step blocks of the packet walk (the blocks with the s_load_dwordx16 of a
    wide node) that exceed the instruction budget.  The walk is bound by what ONE wave issues per node:
    scalar instructions count as much as vector ones, and the compiler has turned the loop's uniform branches into mask
    arithmetic before (a dozen more per node).  Today: 36 VALU + 43..46 others (mixed-sign instantiation: 60 + 47..50); before the diet 37 + 61."""
    out = []
    kernel, block, ins = None, None, []
    def close():
        if kernel and any(i.startswith("s_load_dwordx16") for i in ins):
            nv = sum(i.startswith("v_") for i in ins)
            if nv > max_valu or len(ins) - nv > max_other: out.append((kernel, block, nv, len(ins) - nv))
    for line in asm_text.split("\n"):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            close(); kernel, block, ins = m.group(1), "entry", []
            continue
        if t.startswith(".LBB") or t.startswith("; %bb."):
            close(); block, ins = t.split(":")[0].lstrip("; "), []
            continue
        if t.startswith(".Lfunc_end"):
            close(); kernel, ins = None, []
            continue
        if t and not t.startswith((";", ".")): ins.append(t)
    return out


def main(extra=()):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "kernels.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *makefile_flags(), *extra, "--cuda-device-only", "-S", os.path.join(CSRC, "kernels.hip"), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
        found = find_overlaps(text)
        over = packet_step_budget(text)
    for k, a, b in found:
        print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120])
        print("   ", a)
        print("   ", b)
    print(f"{len(found)} overlapping load pair(s)")
    for k, b, nv, no in over:
        print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120], b, f"{nv} VALU + {no} other instructions")
    print(f"{len(over)} packet node step(s) over the instruction budget")
    return 1 if found or over else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
