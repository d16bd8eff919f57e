#!/usr/bin/env python3
"""Round-6 counter passes on the forward bench (run on the GPU box from the repo root: python3 tools/pmc_r6.py <tag> [g10|g16 ...]):
the vector-memory path of the edge / atom kernels -- texture addresser (TA), texture data (TD), vector L1 (TCP), L2 (TCC) -- and the LDS
bank-conflict counters, one rocprofv3 --pmc pass per group (never next to a tracing domain; the program itself stands after `--`).
This process never touches the GPU: it only starts rocprofv3 children.  A group that rocprofv3 refuses as a whole (too many counters
for one block) is retried block by block.  Output: gpurun_out/pmc_<tag><shape>_<group>/ (read by tools/pmc_report.py <tag><shape>).

Groups hold at most two counters per hardware block (the per-block slot counts of TA / TD / TCP are not documented for gfx950; TCC has
four, SQ eight: MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
GROUPS = {
    "A": [["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"], ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum"], ["TD_TD_BUSY_sum", "TD_TC_STALL_sum"], ["GRBM_GUI_ACTIVE"]],
    "B": [["TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TOTAL_WAVEFRONTS_sum"], ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TD_LOAD_WAVEFRONT_sum", "TD_STORE_WAVEFRONT_sum"]],
    "C": [["TA_FLAT_READ_WAVEFRONTS_sum", "TA_FLAT_WRITE_WAVEFRONTS_sum"], ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TOTAL_ACCESSES_sum"],
          ["TCC_REQ_sum", "TCC_READ_sum"]],
    "D": [["TA_BUSY_avr", "TA_BUSY_max"], ["TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"],
          ["TCC_TAG_STALL_sum", "TCC_BUSY_sum"]],
    "E": [["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum"], ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"], ["TA_ADDR_STALLED_BY_TD_CYCLES_sum"]],
    "L": [["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL",
           "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD"]],
    "S": [["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CU_CYCLES",
           "SQ_ACTIVE_INST_ANY"], ["GRBM_GUI_ACTIVE"]],
}
SHAPES = {"g10": (20, 10), "g16": (64, 16)}  # bench --steps / --warmup: the driver's 10 + 10 batches, four groups of 16


def one_pass(name, counters, steps, warm):
    d = os.path.join(OUT, name)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"), "--no-extras",
           "--steps", str(steps), "--warmup", str(warm), "--min-time", "0.3", "--prewarm", "0.3"]
    with open(d + ".log", "w") as log:
        rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")).returncode
    ok = bool(glob.glob(os.path.join(d, "*", "*counter_collection.csv")))
    print("%s: rc %d, %s (%s)" % (name, rc, "ok" if ok else "NO OUTPUT", " ".join(counters)), flush=True)
    return ok


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    shapes = [s for s in sys.argv[2:] if s in SHAPES] or list(SHAPES)
    groups = [g for g in sys.argv[2:] if g in GROUPS] or list(GROUPS)
    os.makedirs(OUT, exist_ok=True)
    for shape in shapes:
        steps, warm = SHAPES[shape]
        for g in groups:
            blocks = GROUPS[g]
            if not one_pass("pmc_%s%s_%s" % (tag, shape, g), [c for b in blocks for c in b], steps, warm) and len(blocks) > 1:
                for i, b in enumerate(blocks):
                    one_pass("pmc_%s%s_%s%d" % (tag, shape, g, i), b, steps, warm)


if __name__ == "__main__":
    main()
