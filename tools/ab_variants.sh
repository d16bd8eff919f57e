#!/bin/bash
# Builds / switches compared on ONE box (bench.py --no-extras --group $G, default 10).  stdin: lines of
#   label  library-under-scann--material_amd/lib  SCANN_FUSE_LAYERS  SCANN_LF_DELAY
# e.g.  printf "separate libscann_hip.so 0 400\nfused libscann_hip.so 1 400\n" | bash tools/ab_variants.sh
mkdir -p gpurun_out/lf
run() {
  SCANN_HIP_LIB=$PWD/scann--material_amd/lib/$2 SCANN_FUSE_LAYERS=$3 SCANN_LF_DELAY=$4 timeout -k 10 200 python bench.py --no-extras --group ${G:-10} --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-34s %.0f molecules/s  sampled launch %.1f us' % ('$1', d['value'], r['avg_launch_us']))" || exit 1
}
while read -r label lib fuse delay; do
  [ -z "$label" ] && continue
  run "$label" "$lib" "$fuse" "$delay"
done
