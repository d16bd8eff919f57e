#!/bin/bash
# A/B/C... of builds on ONE box, training step (bench.py --train, batch 128 unless BATCH is set):  tools/ab_train5.sh <reps> <lib> [<lib> ...]
reps=$1; shift
batch=${BATCH:-128}
root=${GRAFT_REPO_ROOT:-$PWD}
for r in $(seq $reps); do
  for v in "$@"; do
    ms=$(SCANN_HIP_LIB=$root/scann--material_amd/lib/$v python3 $root/bench.py --train --no-extras --steps 300 --warmup 20 --batch $batch | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "batch $batch $v $ms ms" | tee -a $root/gpurun_out/ab_train5.txt
  done
done
