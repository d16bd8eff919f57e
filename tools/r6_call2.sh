#!/bin/bash
# Round-6 GPU call 2: A/B of the LDS layouts (v0 = round 5: 8-byte plane stores + [row][..] statistics / logits; v1 = 8-byte stores +
# transposed statistics / logits; v2 = 16-byte stores except the gated-row epilogue + transposed; default = 16-byte stores + transposed),
# LDS counters of v1 and the default, the plain backward's readout stage with per-input attribution, the GPU suite.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 420 bash tools/ab5.sh 2 libscann_hip_v0.so libscann_hip_v1.so libscann_hip_v2.so libscann_hip.so > $O/r6_ab_lds.txt 2>&1; echo "ab rc $?"; cat $O/r6_ab_lds.txt
timeout -k 10 300 python3 tools/debug_plain_grads.py 7797 1111 1847 > $O/r6_debug_plain2.txt 2>&1; echo "debug rc $?"; grep "with only" $O/r6_debug_plain2.txt
SCANN_HIP_LIB=$R/scann--material_amd/lib/libscann_hip_v1.so timeout -k 10 200 python3 tools/pmc_r6.py r06v1 g16 g10 L 2>&1 | tee $O/r6_pmc_v1.log
timeout -k 10 200 python3 tools/pmc_r6.py r06c g16 g10 L 2>&1 | tee $O/r6_pmc_c.log
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $O/r6_gpu_suite2.log 2>&1; echo "suite rc $?"; tail -3 $O/r6_gpu_suite2.log
