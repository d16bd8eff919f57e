#!/bin/bash
# A/B/C... of several builds of the library on ONE box, forward bench:  tools/ab_libs.sh <reps> <lib> [<lib> ...]
# (libs relative to scann--material_amd/lib/; boxes differ by ~10 %, builds are only ever compared inside one call)
reps=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
for r in $(seq $reps); do
  for v in "$@"; do
    SCANN_HIP_LIB=$root/scann--material_amd/lib/$v python3 $root/bench.py --no-extras --steps 800 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['roofline']['per_forward_ms']; print('%-28s %.0f molecules/s  edge kernel %.1f us  atom %.3f ms' % ('$v', d['value'], d['roofline']['avg_launch_us'], p['ms_atom']))"
  done
done
