#!/bin/bash
# Round-6 GPU call 4: the plain (generic-width) path after the accuracy changes -- four partial sums per dense output, fp64 GlobalAttention
# scores and pooling-backward scalars: the three named batches, a census sweep, its speed, its tests.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/debug_plain_grads.py 7797 1111 1847 > $O/r6_debug_plain_final.txt 2>&1; echo "debug rc $?"; grep -A9 "parameter gradients" $O/r6_debug_plain_final.txt | grep -v "^ --"; grep "dgq   \|dgk   " $O/r6_debug_plain_final.txt
timeout -k 10 120 python3 tools/generic_rate.py > $O/r6_generic_rate.txt 2>&1; echo "rate rc $?"; tail -8 $O/r6_generic_rate.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -m gpu -x -q -k "width or plain or generic" > $O/r6_gen_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/r6_gen_tests.log
timeout -k 10 420 python3 tests/manual/fuzz_grads.py 380 census=8 > $O/r6_fuzz_census.txt 2>&1; echo "fuzz rc $?"; tail -8 $O/r6_fuzz_census.txt
