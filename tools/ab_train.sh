#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by up to ~10 %): alternating runs of bench.py --train.
#   tools/ab_train.sh [batch] [reps]   compares scann--material_amd/lib/libscann_hip_base.so (A) with libscann_hip.so (B)
batch=${1:-128}; reps=${2:-3}
root=${GRAFT_REPO_ROOT:-$PWD}
for r in $(seq $reps); do
  for v in base cur; do
    lib=$root/scann--material_amd/lib/libscann_hip.so
    [ $v = base ] && lib=$root/scann--material_amd/lib/libscann_hip_base.so
    ms=$(SCANN_HIP_LIB=$lib python3 $root/bench.py --train --no-extras --steps 300 --warmup 20 --batch $batch | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "batch $batch $v $ms ms"
  done
done
