#!/bin/bash
# Round-6 GPU call 5: GPU suite on the current tree, then the forward bench on one and on two streams at the driver's arguments and at the
# defaults (no extras).
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $O/r6_gpu_suite3.log 2>&1; echo "suite rc $?"; tail -3 $O/r6_gpu_suite3.log
for rep in 1 2; do
for st in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --streams $st --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps 20  streams $st  %.0f molecules/s  ms/step %.4f  edge %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
  python3 bench.py --streams $st --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default   streams $st  %.0f molecules/s  ms/step %.4f  edge %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
  python3 bench.py --streams $st --group 8 --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('group 8   streams $st  %.0f molecules/s  ms/step %.4f  edge %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; done 2>&1 | tee $O/r6_streams_ab.txt
