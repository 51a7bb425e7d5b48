"""ISA evidence for the node step and the triangle block of the ray-buffer kernel (profiles/<tag>_isa.txt).

Compiles kernels.hip for gfx950 to assembly (device only), takes k_trace_strip<bvh4q, closest, no counters, capped stack>
and reports: instruction totals of the kernel, and per-class histograms + full listing of the two hot basic blocks
(the BVH4q node step: the block with the four node loads; the triangle block: the block with the three triangle loads)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp".split()     # = CXXFLAGS + KFLAGS of the Makefile
KERNEL = "_ZN2rt12_GLOBAL__N_113k_trace_stripILi3ELb0ELb0ELb1ELb1EEE"     # <bvh4q, closest hit, no counters, capped stack, ray queue>

def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "kernels.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "--cuda-device-only", "-S", os.path.join(CSRC, "kernels.hip"), "-o", out], check=True)
        text = open(out).read()
    m = re.search(r"^(%s[^:\n]*):[^\n]*\n(.*?)^\.Lfunc_end" % KERNEL, text, re.S | re.M)
    body = m.group(2).split("\n")
    # basic blocks: a label or "; %bb.N:" starts one
    blocks, cur, name = [], [], "entry"
    for l in body:
        s = l.strip()
        if s.startswith(".LBB") or s.startswith("; %bb."):
            if cur: blocks.append((name, cur))
            cur, name = [], s.split(":")[0].lstrip("; ")
            continue
        if s and not s.startswith(";") and not s.startswith("."): cur.append(s.split(";")[0].strip())
    if cur: blocks.append((name, cur))
    ops = lambda ins: collections.Counter(i.split()[0] for i in ins)
    def classes(ins):
        c = collections.Counter()
        for i in ins:
            o = i.split()[0]
            c["VALU" if o.startswith("v_") else "SALU" if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_barrier")) else
              "VMEM" if o.startswith(("global_", "flat_", "buffer_", "scratch_")) else "LDS" if o.startswith("ds_") else "branch" if o.startswith(("s_cbranch", "s_branch")) else "wait/nop/other"] += 1
        return dict(c)
    allins = [i for _, b in blocks for i in b]
    print("kernel k_trace_strip<bvh4q, closest, no counters, capped stack, ray queue>: %d instructions in %d basic blocks" % (len(allins), len(blocks)))
    print("  by class:", classes(allins))
    res = re.search(r"^\s*\.vgpr_count:\s*(\d+)", text[m.end():m.end() + 200000], re.M)
    node = max(blocks, key=lambda b: sum(i.startswith("v_cvt_f32_ubyte") for i in b[1]))
    loads = lambda b: sum(i.startswith(("global_load_dwordx3", "global_load_dwordx4", "flat_load_dwordx3", "flat_load_dwordx4")) for i in b[1])
    tri = max((b for b in blocks if b is not node and loads(b) == 3), key=lambda b: sum(i.startswith("v_") for i in b[1]))
    for title, (nm, ins) in (("BVH4q node step (four children: decode, slab test, sort; the pushes follow in the next blocks)", node), ("triangle block (one Moeller-Trumbore test per lane; the accept path -- a few compares and selects -- follows in the next blocks)", tri)):
        print("\n== %s: block %s, %d instructions ==" % (title, nm, len(ins)))
        print("  by class:", classes(ins))
        print("  by opcode:", ", ".join("%s x%d" % kv for kv in sorted(ops(ins).items(), key=lambda kv: -kv[1])))
        for i in ins: print("    " + i)
    packet_report(text)

def packet_report(text):
    """node step of the packet walk (k_primary_packet<no counters, no shading>): the first block that fetches a wide node"""
    m = re.search(r"^(_ZN2rt12_GLOBAL__N_116k_primary_packetILb0ELb0EEE[^:\n]*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    if not m: return
    blocks, cur, name = [], [], "entry"
    for l in m.group(2).split("\n"):
        s_ = l.strip()
        if s_.startswith(".LBB") or s_.startswith("; %bb."):
            if cur: blocks.append((name, cur))
            cur, name = [], s_.split(":")[0].lstrip("; ")
            continue
        if s_ and not s_.startswith(";") and not s_.startswith("."): cur.append(s_.split(";")[0].strip())
    if cur: blocks.append((name, cur))
    cand = [b for b in blocks if any(i.startswith("s_load_dwordx16") for i in b[1]) and any(i.startswith("v_pk_fma") for i in b[1])]
    if not cand: return
    nm, ins = min(cand, key=lambda b: sum(i.startswith("v_") for i in b[1]))     # an octant-specialised copy (the generic one has 24 more min/max)
    valu = [i for i in ins if i.startswith("v_")]
    print("\n== packet walk, node step of one direction octant (k_primary_packet<no counters, no shading>): block %s, %d instructions, %d VALU "
          "(the scalar child ordering and the pushes follow in the next blocks) ==" % (nm, len(ins), len(valu)))
    c = collections.Counter(i.split()[0] for i in ins)
    print("  by opcode:", ", ".join("%s x%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))
    for i in ins: print("    " + i)


if __name__ == "__main__":
    main()
