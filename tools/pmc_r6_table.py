#!/usr/bin/env python3
"""The vector-memory path of the dominant kernels as ONE table (profiles/r06_memory_path.txt), from the per-launch counter averages that
tools/pmc_report.py wrote for the passes of tools/pmc_r6.py:  python3 tools/pmc_r6_table.py r06fg10 r06fg16 [> profiles/r06_memory_path.txt]

Units (checked against this kernel's known byte counts, profiles/r05_notes.md): TCC_EA0_RDREQ are 128-B requests on gfx950 (RDREQ_32B
= 0; 1.727 M x 128 B = 221 MB = the fetch the FETCH_SIZE passes give after the guide's x 2), TCC_EA0_WRREQ are 64-B requests (2.506 M x
64 B = 160 MB = geometry out + context out), TCP_TCC_WRITE_REQ likewise 64 B (2.64 M x 64 B = 169 MB); TCP_TCC_READ_REQ are 128-B lines:
the atom kernel streams 5 x 64 KB of weights per tile with no reuse inside a workgroup (260 MB of loads per launch at the driver's shape)
and sends 1.945 M read requests = 249 MB at 128 B -- at 64 B half of those loads would have to hit a 32 KB L1 that holds a tenth of one
tile's weights.  GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles = value / 8.  *_sum counters are summed
over the 256 CUs (TA / TD / TCP) or the 128 L2 channels (TCC): the table divides by the instance count to get cycles per instance."""
import re
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [("edge_kernel<true, 2, false, false, false, false>", "edge kernel (layers 2..L-1)"),
           ("edge_kernel<true, 2, true, false, false, false>", "edge kernel, first layer (basis MLP fused)"),
           ("atom_kernel<true, 0,", "atom kernel (ResidualNorm + P1 / P3 / q)"),
           ("readout_kernel", "readout kernel")]


def load(tag):
    t = open(os.path.join(ROOT, "profiles", "%s_counters.txt" % tag)).read()
    out = {}
    for b in re.split(r"\n(?=scann::)", t):
        lines = b.strip().split("\n")
        if not lines or not lines[0].startswith("scann::"):
            continue
        out[lines[0].strip()] = {ln.split()[0]: float(ln.split()[1]) for ln in lines[1:] if ln.strip()}
    return out


def main():
    for tag in sys.argv[1:] or ["r06fg10", "r06fg16"]:
        ks = load(tag)
        print("==== %s (rocprofv3 --pmc passes of tools/pmc_r6.py; per-launch averages; 256 CUs, 8 XCDs, 128 L2 channels)" % tag)
        for pat, label in KERNELS:
            name = next((k for k in ks if pat in k), None)
            if not name or "GRBM_GUI_ACTIVE" not in ks[name]:
                continue
            d = ks[name]
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            g = lambda k, n=256.0: d.get(k, float("nan")) / n  # noqa: E731
            rd_instr, wr_instr = d.get("TA_FLAT_READ_WAVEFRONTS_sum", float("nan")), d.get("TA_FLAT_WRITE_WAVEFRONTS_sum", float("nan"))
            l1_bytes = (rd_instr + wr_instr) * 1024.0  # upper bound: every vector-memory wave-instruction a 16-B-per-lane access
            l2_rd = d.get("TCP_TCC_READ_REQ_sum", float("nan")) * 128.0
            l2_wr = d.get("TCP_TCC_WRITE_REQ_sum", float("nan")) * 64.0
            ea_rd, ea_wr = d.get("TCC_EA0_RDREQ_sum", float("nan")) * 128.0, d.get("TCC_EA0_WRREQ_sum", float("nan")) * 64.0
            hit = d.get("TCC_HIT_sum", 0.0) / max(1.0, d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0))
            print("%s\n   %s" % (label, name))
            print("   kernel duration (under the profiler)        %9.0f shader cycles" % cyc)
            print("   vector-memory wave-instructions              %9.0f loads + %.0f stores -> <= %.0f MB through the CU's L1 path = %.1f B / clk / CU"
                  % (rd_instr, wr_instr, l1_bytes / 1e6, l1_bytes / 256.0 / cyc))
            print("   texture addresser busy (TA_TA_BUSY / CU)      %5.1f %% of the cycles (busiest CU %.1f %%); stalled by the cache on addresses %.1f %%, on data %.1f %%"
                  % (100 * g("TA_TA_BUSY_sum") / cyc, 100 * d.get("TA_BUSY_max", float("nan")) / cyc, 100 * g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / cyc,
                     100 * g("TA_DATA_STALLED_BY_TC_CYCLES_sum") / cyc))
            print("   texture data busy (TD_TD_BUSY / CU)           %5.1f %%; of it waiting for the L1's data (TD_TC_STALL) %.1f %% of the cycles"
                  % (100 * g("TD_TD_BUSY_sum") / cyc, 100 * g("TD_TC_STALL_sum") / cyc))
            print("   L1 (TCP): clocks on %.1f %%; stalled on its limit of pending misses %.1f %%, on tag conflicts of reads %.1f %%, on the return path %.1f %%"
                  % (100 * g("TCP_GATE_EN1_sum") / cyc, 100 * g("TCP_PENDING_STALL_CYCLES_sum") / cyc, 100 * g("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum") / cyc,
                     100 * g("TCP_TCR_TCP_STALL_CYCLES_sum") / cyc))
            print("   L1 -> L2 requests                             reads %.0f MB (%.0f requests x 128 B), writes %.0f MB: the L1 serves ~%.0f %% of the loaded bytes itself"
                  % (l2_rd / 1e6, d.get("TCP_TCC_READ_REQ_sum", float("nan")), l2_wr / 1e6, 100 * max(0.0, 1 - l2_rd / max(1.0, rd_instr * 1024.0))))
            print("   L2 read bandwidth used                        %.1f TB/s of ~34.5 (MI355X_MICROARCH.md, L2)" % (l2_rd / (cyc / 2.2e9) / 1e12))
            print("   L2 (TCC): hit rate %.1f %%, busy %.1f %% of the cycles per channel, tag stalls %.2f %%; beyond the L2 (Infinity Cache / HBM): %.0f MB read + %.0f MB written"
                  % (100 * hit, 100 * g("TCC_BUSY_sum", 128.0) / cyc, 100 * g("TCC_TAG_STALL_sum", 128.0) / cyc, ea_rd / 1e6, ea_wr / 1e6))
            if "SQ_LDS_IDX_ACTIVE" in d:
                print("   LDS: %.0f instructions, bank-conflict cycles %.0f of %.0f active = %.1f %%"
                      % (d["SQ_INSTS_LDS"], d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"], 100 * d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]))
            if "SQ_WAVE_CYCLES" in d:
                print("   waves: issuing %.1f %%, waiting for an instruction's issue %.1f %%, parked (waitcnt / barrier) %.1f %% of their lifetime; MFMA pipe busy %.1f %% of the CU cycles"
                      % (100 * d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                         100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / max(1.0, d["SQ_BUSY_CU_CYCLES"])))
        print()


if __name__ == "__main__":
    main()
