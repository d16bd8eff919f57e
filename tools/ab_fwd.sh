#!/bin/bash
# A/B of two builds of the library on ONE box, forward bench:  tools/ab_fwd.sh [reps] [extra bench args]
reps=${1:-3}; shift
root=${GRAFT_REPO_ROOT:-$PWD}
for r in $(seq $reps); do
  for v in base cur; do
    lib=$root/scann--material_amd/lib/libscann_hip.so
    [ $v = base ] && lib=$root/scann--material_amd/lib/libscann_hip_base.so
    SCANN_HIP_LIB=$lib python3 $root/bench.py --no-extras --steps 800 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v %.0f molecules/s  edge kernel %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
  done
done
