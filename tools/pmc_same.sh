#!/bin/bash
# dynamic instruction mix of k_qp_solve on 256 identical problems (run on the GPU box: gpurun -- bash tools/pmc_same.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/same256.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np
from trajopt_amd import configs, abi, runtime
pci, s, g = configs.config1()
ctx = runtime.Context(0)
ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
one = configs.seeds_for(1, pci, s, g, 1)
x0 = np.repeat(one, 256, axis=0)
ctx.set_x0(x0); ctx.convexify()
xq, cvx, rec = ctx.qp_solve()
print("iters", rec[0].osqp_iter)
PY
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc_same/$tag -o out --output-format csv -- python /tmp/same256.py > /tmp/log_$tag.txt 2>&1
  tail -2 /tmp/log_$tag.txt
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmc_same/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_qp_solve" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:40s} {v:.4g}")
PY
