for q in 4 6 8 16; do for ns in 2 3 4; do
  f=$(GPU_MAX_HW_QUEUES=$q SCANN_STREAMS=$ns timeout -k 10 100 python3 tools/fit_rate.py 2>&1 | tail -1 | sed -E 's/.*\(([0-9.]+) ms\/step.*/\1/')
  e=$(GPU_MAX_HW_QUEUES=$q SCANN_STREAMS=$ns timeout -k 10 100 python3 tools/e2e_breakdown.py 8 2>&1 | grep "resident, forward" | sed -E 's/.*only: ([0-9]+) mol.*/\1/')
  b=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 100 python3 bench.py --train --no-extras --steps 200 --warmup 20 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))")
  echo "queues=$q streams=$ns fit_ms=$f resident_fwd=$e train_ms=$b"
done; done
