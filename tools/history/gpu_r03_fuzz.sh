#!/bin/bash
# larger slices of the randomised parity sweep on the device for the round-3 term families (the tiers run 12 - 16 cases)
O=gpurun_out/r03fz; mkdir -p $O
timeout 400 python tests/tools/fuzz_parity.py 40 23 gpu kin > $O/fuzz_kin_40.log 2>&1; tail -2 $O/fuzz_kin_40.log
