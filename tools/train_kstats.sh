#!/bin/bash
# per-kernel medians of one training configuration under rocprofv3 (kernel trace only):  tools/train_kstats.sh <tag> [batch] [lib]
tag=${1:-t}; batch=${2:-128}; lib=${3:-}
root=${GRAFT_REPO_ROOT:-$PWD}
[ -n "$lib" ] && export SCANN_HIP_LIB=$root/scann--material_amd/lib/$lib
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/ks_$tag
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/ks_$tag -- python3 $root/bench.py --train --no-extras --steps 200 --warmup 20 --batch $batch > $root/gpurun_out/ks_$tag.log 2>&1
python3 - $root/gpurun_out/ks_$tag <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
n_adam = 0
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('scann::', '')
    d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    n_adam += 'adam_kernel' in n
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / n_adam > 8:
        print("%-30s n/step=%5.1f med=%7.2f us  per step %7.1f" % (k[:30], len(v) / n_adam, statistics.median(v), sum(v) / n_adam))
    tot += sum(v) / n_adam
print("sum of kernel durations per step %.1f us" % tot)
PY
