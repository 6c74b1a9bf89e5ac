#!/usr/bin/env python3
"""Code-object metadata of the library's kernels (register counts, spills, private segment, LDS): python tools/kernel_meta.py lib.so [kernel substring]
Reads the .note section (msgpack metadata) through llvm-readelf --notes."""
import re, subprocess, sys
lib = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_sqp"
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
# the device code object is embedded in the host .so (.hip_fatbin): extract with clang-offload-bundler
import tempfile, os
tmp = tempfile.mkdtemp()
co = os.path.join(tmp, "dev.co")
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "discard.so")])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
txt = subprocess.check_output([readelf, "--notes", co], text=True)
cur = {}
for line in txt.split("\n"):
    m = re.match(r"\s*-?\s*\.?(\w[\w.]*):\s*(.*)$", line.strip().lstrip("- "))
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        pass
    if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "name", "symbol"):
        cur[k.lstrip(".")] = v
    if k == "wavefront_size" or k == "workgroup_processor_mode":
        pass
    if k == "vgpr_spill_count":
        if pat in cur.get("name", ""):
            print(f"{cur.get('name','?')[:60]:60s} vgpr {cur.get('vgpr_count')} agpr {cur.get('agpr_count')} sgpr {cur.get('sgpr_count')} "
                  f"vgpr_spills {cur.get('vgpr_spill_count')} sgpr_spills {cur.get('sgpr_spill_count')} private_segment {cur.get('private_segment_fixed_size')} B")
        cur = {}
