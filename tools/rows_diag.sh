for d in 0 1 2 3 4 8 15; do SCANN_ROWS_DIAG=$d python bench.py --no-extras --min-time 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('diag $d', round(d['value']), d['roofline']['avg_launch_us'])"; done
