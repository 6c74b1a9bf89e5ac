#!/usr/bin/env python3
"""List the loops (backward branches) of one kernel in a hipcc -S dump with an instruction-class histogram each."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern) and l.rstrip().split(":")[0] == l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
ins = []  # (idx_in_ins, text)
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    ins.append(t.split(";")[0].strip())
def cls(t):
    op = t.split()[0]
    if op.startswith("v_fma_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64") or op.startswith("v_max_f64") or op.startswith("v_min_f64"): return "valu_f64"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_div") or op.startswith("v_rcp") or op.startswith("v_sqrt") or op.startswith("v_rsq"): return "valu_f64_slow"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_nop"): return "nop"
    return "salu"
loops = []
for i, t in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= i:
            loops.append((labels[lab], i, lab))
print(f"{kern}: {len(ins)} instructions, {len(loops)} backward branches")
for a, b, lab in sorted(loops, key=lambda x: -(x[1] - x[0])):
    n = b - a + 1
    if n < minlen:
        continue
    h = collections.Counter(cls(t) for t in ins[a:b + 1])
    print(f"  loop {lab} [{a}..{b}] {n} instr: " + ", ".join(f"{k} {v}" for k, v in h.most_common()))
