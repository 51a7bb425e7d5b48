"""Static check of the generated gfx950 code: within one group of vector-memory loads that are in flight together (no
s_waitcnt vmcnt between them), no load may write registers that an earlier load of the group writes.  The register
allocator produces this when part of a wide load is dead (it hands the dead registers to the next load); the hardware
then cannot overlap the two loads, which cost the out-of-cache configurations 12 % (profiles/r02s_c4_regression.txt).

usage: check_load_overlap.py [extra hipcc flags]      exit status 1 if an overlap is found in a traversal kernel"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def makefile_flags():
    text = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    return flags


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


def main(extra=()):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "kernels.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *makefile_flags(), *extra, "--cuda-device-only", "-S", os.path.join(CSRC, "kernels.hip"), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        found = find_overlaps(open(asm).read())
    for k, a, b in found:
        print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120])
        print("   ", a)
        print("   ", b)
    print(f"{len(found)} overlapping load pair(s)")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
