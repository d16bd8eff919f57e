#!/bin/bash
# Round-5 A/B of builds on ONE box: same bytes? then the forward bench at the driver's 10-batch shape and at 16 batches per launch.
#   tools/ab5.sh <reps> <lib> [<lib> ...]      (libs relative to scann--material_amd/lib/; output also in gpurun_out/ab5.txt)
reps=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/ab5.txt
mkdir -p $root/gpurun_out
libs=()
for v in "$@"; do libs+=("$root/scann--material_amd/lib/$v"); done
python3 $root/tools/ab_bits.py "${libs[@]}" 2>&1 | tee -a $out
for r in $(seq $reps); do
  for v in "$@"; do
    for shape in "--steps 200 --group 10" "--steps 800"; do
      SCANN_HIP_LIB=$root/scann--material_amd/lib/$v python3 $root/bench.py --no-extras $shape | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['roofline']['per_forward_ms']; print('%-28s %-24s %.0f molecules/s  edge kernel %.1f us  atom %.3f ms  frac %.3f' % ('$v', '$shape', d['value'], d['roofline']['avg_launch_us'], p['ms_atom'], d['roofline']['frac']))" 2>&1 | tee -a $out
    done
  done
done
