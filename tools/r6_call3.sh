#!/bin/bash
# Round-6 GPU call 3: does the accuracy of the plain dense kernel's sums decide the plain backward's distance from fp64 on the batches
# the review named?  default (serial fp32 chains) / four interleaved partial sums / exact products with fp64 sums, same three batches,
# then a short randomised sweep of each (tests/manual/fuzz_grads.py).
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in "" _d4 _d64; do
  SCANN_HIP_LIB=$R/scann--material_amd/lib/libscann_hip$v.so timeout -k 10 200 python3 tools/debug_plain_grads.py 7797 1111 1847 > $O/r6_debug_dense$v.txt 2>&1
  echo "== dense$v rc $?"; grep -A9 "parameter gradients" $O/r6_debug_dense$v.txt | grep -v "^ --"
done
for v in "" _d4 _d64; do
  SCANN_HIP_LIB=$R/scann--material_amd/lib/libscann_hip$v.so timeout -k 10 260 python3 tests/manual/fuzz_grads.py 150 > $O/r6_fuzz_dense$v.txt 2>&1
  echo "== fuzz dense$v rc $?"; tail -3 $O/r6_fuzz_dense$v.txt; grep -c "ill-conditioned" $O/r6_fuzz_dense$v.txt
done
