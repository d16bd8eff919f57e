#!/bin/bash
# Round-6 GPU call 1 (from the repo root on the GPU box): GPU suite on the paired plane stores, A/B against the 8-byte-store build,
# the plain-fp32 backward's readout stage on the fuzz batches the review named, TA / TD / TCP / TCC / LDS counter passes.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $O/r6_gpu_suite1.log 2>&1; echo "suite rc $?"; tail -2 $O/r6_gpu_suite1.log
timeout -k 10 300 bash tools/ab5.sh 2 libscann_hip_b64.so libscann_hip.so > $O/r6_ab_b128.txt 2>&1; echo "ab rc $?"; cat $O/r6_ab_b128.txt
timeout -k 10 300 python3 tools/debug_plain_grads.py 1111 7797 1847 > $O/r6_debug_plain.txt 2>&1; echo "debug rc $?"; tail -5 $O/r6_debug_plain.txt
timeout -k 10 600 python3 tools/pmc_r6.py r06 2>&1 | tee $O/r6_pmc.log
