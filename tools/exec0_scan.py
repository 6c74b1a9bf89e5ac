#!/usr/bin/env python3
"""Signature of a control-flow miscompile in a gfx950 code object: VECTOR instructions that can only ever execute with EXEC = 0.

A block whose every predecessor edge is an `s_cbranch_execz` (taken only when no lane is active) and that has no fall-through
predecessor runs with EXEC = 0 until something restores EXEC (`s_or_b64 exec, exec, ...`, `s_mov_b64 exec, ...`).  A vector
instruction there (v_mov_b32 v21, v1 ...) is a no-op for every lane.  The compiler never emits such code on purpose: it is what is
left when the END_CF of an inner divergent region was merged into the END_CF of the enclosing one and a register copy (PHI
elimination / live-range split) later landed between the two.  The value the copy was meant to define is then STALE in every lane
once EXEC is restored (round 5: the zero voffset of a global_load in k_sqp_pool).

  python tools/exec0_scan.py lib.so|code_object.co [more ...]     exit code 1 if any site is found
"""
import os, re, shutil, subprocess, sys, tempfile


def llvm_bin():
    """the LLVM tools of the ROCm toolchain that built the library: $TMX_LLVM_BIN, else beside $HIPCC / $ROCM_PATH, else /opt/rocm"""
    cands = []
    if os.environ.get("TMX_LLVM_BIN"):
        cands.append(os.environ["TMX_LLVM_BIN"])
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    raise SystemExit("exec0_scan: no llvm-objdump found (set TMX_LLVM_BIN)")


LLVM = llvm_bin()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, tmp):
    """every gfx950 code object of the library: one offload bundle per translation unit in .hip_fatbin"""
    if path.endswith(".co") or path.endswith(".s"):
        return [path]
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(tmp, "discard.so")])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(blob)
        part, co = os.path.join(tmp, f"bundle{k}.bin"), os.path.join(tmp, f"dev{k}.co")
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
        out.append(co)
    return out


def disasm(path):
    if path.endswith(".s"):
        return open(path).read().split("\n")
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", path], text=True).split("\n")


INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
TGT = re.compile(r"<([^>+]+)(?:\+0x([0-9a-f]+))?>\s*$")
FUNC = re.compile(r"^([0-9a-f]+) <(.*)>:")


def scan(path):
    """every code object of the library on its own (the objects of two translation units overlap in their addresses); returns the
    sites and the names of the functions looked at"""
    sites, names = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(path, tmp):
            s, n = scan_lines(disasm(co))
            sites += s
            names += n
    return sites, names


def scan_lines(lines):
    funcs, cur = [], None
    for ln in lines:
        m = FUNC.match(ln)
        if m:
            cur = {"name": m.group(2), "base": int(m.group(1), 16), "ins": []}
            funcs.append(cur)
            continue
        m = INS.match(ln)
        if m and cur is not None:
            cur["ins"].append((int(m.group(3), 16), m.group(1), m.group(2), ln))
    sites = []
    for f in funcs:
        ins = f["ins"]
        index = {a: i for i, (a, _, _, _) in enumerate(ins)}
        preds = {}  # target address -> list of branch mnemonics
        for a, op, args, ln in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                m = TGT.search(ln)
                if m:
                    preds.setdefault(f["base"] + int(m.group(2) or "0", 16), []).append(op)
        for t, ops in preds.items():
            if t not in index or any(o != "s_cbranch_execz" for o in ops):
                continue
            i = index[t]
            if i > 0 and ins[i - 1][1] not in ("s_branch", "s_endpgm", "s_setpc_b64"):
                continue  # fall-through predecessor: EXEC may be non-zero
            # branch relaxation: `s_cbranch_execnz FAR` becomes `s_cbranch_execz L; s_getpc / s_add / s_addc / s_setpc FAR; L:` - L is
            # the EXEC = 0 side of an edge block (spill code of a split critical edge: nothing to save, nothing lost)
            if i >= 5 and [x[1] for x in ins[i - 5:i]] == ["s_cbranch_execz", "s_getpc_b64", "s_add_u32", "s_addc_u32", "s_setpc_b64"] and len(ops) == 1:
                continue
            j = i
            while j < len(ins):
                a, op, args, ln = ins[j]
                if j > i and a in preds:
                    break  # another label: other predecessors join
                if re.match(r"s_(or|mov|and|andn2|xor|or_saveexec|and_saveexec)\w*\s", op + " ") and re.match(r"exec\b", args):
                    break
                if op.startswith("s_or_saveexec") or op.startswith("s_and_saveexec"):
                    break
                if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if re.match(r"(v_|ds_|global_|flat_|scratch_|buffer_)", op) and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                    sites.append((f["name"], a - f["base"], op + " " + args))
                j += 1
    return sites, [f["name"] for f in funcs]


def main():
    bad = 0
    for p in sys.argv[1:]:
        s, names = scan(p)
        kernels = set()
        for n in names:
            m = re.match(r"_Z(\d+)(k_\w+)", n)  # Itanium: _Z <length> <name> <parameter types>
            if m:
                kernels.add(m.group(2)[:int(m.group(1))])
            elif n.startswith("k_"):
                kernels.add(n)
        kernels = sorted(kernels)
        print(f"{p}: {len(names)} function(s) scanned, kernels: " + " ".join(kernels))
        print(f"{p}: {len(s)} vector instruction(s) that can only run with EXEC = 0")
        for name, off, txt in s:
            print(f"   {name[:60]}+0x{off:x}: {txt}")
        bad += len(s)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
