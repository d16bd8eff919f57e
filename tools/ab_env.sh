#!/bin/bash
# A/B of an environment switch on one box, training step (bench.py --train, batch 128):  tools/ab_env.sh <reps> <VAR> <value> [<value> ...]
# ("-" as a value: the variable unset)
reps=$1; var=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
rm -f $root/gpurun_out/ab_env.txt
for r in $(seq $reps); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    ms=$(python3 $root/bench.py --train --no-extras --steps 300 --warmup 20 --batch ${BATCH:-128} | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$var=$v $ms" >> $root/gpurun_out/ab_env.txt
  done
done
python3 - <<PY
import collections,statistics
d=collections.defaultdict(list)
for l in open("$root/gpurun_out/ab_env.txt"):
    p=l.split(); d[p[0]].append(float(p[1]))
for k,v in d.items(): print(k, "median %.4f mean %.4f min %.4f n=%d"%(statistics.median(v), statistics.mean(v), min(v), len(v)))
PY
