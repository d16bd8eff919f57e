"""One process per GPU without an external launcher.

``spawn_ranks(argv, n)`` starts ``n`` fresh Python processes running ``argv`` with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set (the variables ``torch.distributed.run`` would set, so a script behaves the same under
either launcher) and waits for them.  The parent makes NO HIP call and does not load ``libscann_hip.so``: a process
that has initialised the GPU must never be the one that forks or execs the ranks.
"""
from __future__ import annotations

import os
import secrets
import socket
import subprocess
import sys
import time
import uuid


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv, n, env=None, timeout=None):
    """Run ``[sys.executable] + argv`` as ranks 0..n-1 (rank r on device r).  stdout / stderr of every rank are inherited,
    so rank 0's report reaches the caller's stdout.  Returns the largest exit code."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                SCANN_RDZV_ID=uuid.uuid4().hex,
                SCANN_RDZV_SECRET=secrets.token_hex(32))  # per-job secret of the ranks' TCP rendezvous (never on the command line)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    rc, t0 = 0, time.time()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = max(rc, abs(code))
            if rc or (timeout is not None and time.time() - t0 > timeout):
                rc = rc or 124
                break  # a rank that failed must not leave its siblings waiting in a barrier
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc
