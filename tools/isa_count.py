#!/usr/bin/env python3
"""tools/isa_count.py KERNEL_SUBSTRING [asm]: static tally of one kernel of build/wfpt_kernels.s (tools/isa_dump.sh) -- instructions per class
for the whole kernel and for every innermost loop that holds LDS node-pair reads (the traversal's inner visit), plus registers and spills.
Writes the kernel's text to build/<name>.s for reading."""
import re
import sys

needle = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "build/wfpt_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and needle in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
open("build/kernel_" + re.sub(r"\W", "_", needle)[:60] + ".s", "w").write("\n".join(body))


def cls(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "BRANCH"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio"):
        return "MISC"
    if op.startswith("s_load") or op.startswith("s_store") or op.startswith("s_buffer") or op.startswith("s_memtime") or op.startswith("s_dcache"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    return "OTHER"


def tally(seg):
    t = {}
    for l in seg:
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m:
            k = cls(m.group(1))
            t[k] = t.get(k, 0) + 1
    return t


print(lines[start].rstrip(":"))
print("whole kernel:", tally(body))
for l in lines[end:end + 80]:
    if re.search(r"\.(sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size)|NumSgprs|NumVgprs|ScratchSize|Occupancy", l):
        print("   ", l.strip())
# innermost loops: a label .LBBx_y ... a backward branch to it, without another backward branch inside
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\t(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))
for a, b in loops:
    inner = [x for x in loops if x != (a, b) and a <= x[0] and x[1] <= b]
    seg = body[a:b + 1]
    n_ds128 = sum(1 for l in seg if "ds_read_b128" in l)
    if n_ds128 >= 4 and len(seg) < 400:
        t = tally(seg)
        print(f"loop lines {a}-{b} ({len(inner)} loops nested inside): {t}")
