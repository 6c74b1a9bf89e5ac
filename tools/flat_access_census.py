#!/usr/bin/env python3
"""16-byte FLAT accesses of the device code, by kernel and source line.

    make -C trajopt_amd/csrc OUT=/tmp/census EXTRA=-gline-tables-only        # same code, with line tables
    python tools/flat_access_census.py /tmp/census/libtrajopt_mi355x.so [kernel-name-substring ...]

The backend merges adjacent 8-byte loads through a GENERIC pointer into flat_load_dwordx4 at 4-byte alignment.  In HBM that is legal
at any alignment; when the pointer is an LDS array whose rows start on 8-byte boundaries (odd block sizes) it is the access that has
produced this project's memory-aperture violations and the device-only failure analysed in
profiles/r04/r04_defect_offline_analysis.md.  Typed LDS pointers get ds_read2_b64 instead.  The listing names the lines to look at;
whether a line indexes an LDS-resident array is for the reader to decide.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
WIDE = re.compile(r"\b(flat_load_dwordx[34]|flat_store_dwordx[34])\b")


def device_code_object(lib, workdir):
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    targets = subprocess.check_output([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fat], text=True).split()
    gfx = [t for t in targets if "gfx950" in t]
    if not gfx:
        sys.exit("no gfx950 code object in " + lib)
    co = os.path.join(workdir, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=" + gfx[0], "--output=" + co])
    return co


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    lib, wanted = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        co = device_code_object(lib, d)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-l", co], capture_output=True, text=True, check=True).stdout
    sites = collections.Counter()
    fn, loc = "?", "?"
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            fn = m.group(1)
            continue
        if line.startswith("; "):
            loc = os.path.relpath(line[2:].strip(), os.getcwd()) if line[2:].startswith("/") else line[2:].strip()
            continue
        m = WIDE.search(line)
        if m and (not wanted or any(w in fn for w in wanted)):
            sites[(fn, loc, m.group(1))] += 1
    try:
        names = subprocess.run(["c++filt"], input="\n".join(sorted({k[0] for k in sites})), capture_output=True, text=True).stdout.split("\n")
        pretty = dict(zip(sorted({k[0] for k in sites}), names))
    except OSError:
        pretty = {}
    last = None
    for (f, l, op), n in sorted(sites.items()):
        if f != last:
            print(pretty.get(f, f)[:150])
            last = f
        print(f"  {n:4d} x {op:20s} {l}")
    print(f"{sum(sites.values())} wide flat accesses in {len({k[0] for k in sites})} functions")


if __name__ == "__main__":
    main()
