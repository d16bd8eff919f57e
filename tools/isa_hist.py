#!/usr/bin/env python3
"""Per-kernel instruction histogram of the gfx950 ISA hipcc makes of a .hip file (CPU only: cross-compiles, needs no GPU).

    python tools/isa_hist.py                                  # scann_kernels.hip, every kernel, one line each
    python tools/isa_hist.py -k 'edge_kernel<true, 2, false, false>' -v   # one kernel, opcode-by-opcode
    python tools/isa_hist.py --src scann_train_fused.hip -D SCANN_STAMPS

Counts are STATIC wave-instructions of the kernel's text.  The streamed forward kernels are fully unrolled and their only loops are the
softmax / context loops over an atom's edges, so for them static count ~ executed count per tile-wave outside those loops (PMC check of
round 4: SQ_INSTS_VALU / tile-waves = 1,313 against a static 1,315).  Classes:

  mfma      v_mfma_*                                matrix pipe
  valu      every other v_* instruction             vector ALU -- broken down into
    move      v_mov_b32 / v_accvgpr_* / v_readlane ... register moves
    select    v_cndmask_b32                         selects (ragged-tail guards, clamps)
    addr      integer add / shift / mul / mad / bit ops (address and index arithmetic, mostly)
    cvt       conversions (the hi / lo fp16 split: v_cvt_pk_f16_f32, v_fma_mix_*)
    trans     quarter-rate transcendentals (v_exp, v_rcp, v_rsq, v_sqrt, v_log)
    cmp       v_cmp_* / v_cmpx_*
    xlane     DPP / permlane / v_readfirstlane cross-lane moves that are VALU-issued
    fp        the rest: the model's fp32 arithmetic (v_fma, v_mul, v_add, v_sub, v_max ...)
  lds       ds_*                                    (ds_bpermute / ds_swizzle included)
  vmem      global_* / buffer_* / scratch_* / flat_*  loads | stores
  salu      s_* except the ones below
  branch    s_cbranch_* / s_branch, and s_and_saveexec-style exec-mask edits (what a per-lane `if` costs)
  wait      s_waitcnt / s_nop / s_barrier / s_sleep
"""
import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scann--material_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFILT = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")
MOVE = ("v_mov_b32", "v_mov_b64", "v_accvgpr_", "v_swap_b32", "v_pk_mov_b32")
XLANE = ("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute", "v_mov_b32_dpp", "v_add_f32_dpp", "v_max_f32_dpp")
ADDR = ("v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_sub_co", "v_subrev_u32", "v_subrev_co", "v_lshl", "v_lshr", "v_ashr", "v_mul_lo", "v_mul_hi",
        "v_mul_u32", "v_mul_i32", "v_mad_u", "v_mad_i", "v_and_b32", "v_or_b32", "v_xor_b32", "v_and_or", "v_or3", "v_bfe", "v_bfi", "v_add3_u32",
        "v_add_lshl", "v_min_u", "v_min_i", "v_max_u", "v_max_i", "v_med3_i", "v_med3_u", "v_mbcnt", "v_not_b32", "v_add_nc_u32", "v_sub_nc_u32",
        "v_lshl_add", "v_lshl_or", "v_xad_u32", "v_perm_b32", "v_alignbit", "v_add_i32", "v_sub_i32")
CVT = ("v_cvt_", "v_fma_mix", "v_pack_b32")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma", None
    if op.startswith("v_"):
        if "_dpp" in op or op.startswith(XLANE):
            return "valu", "xlane"
        if op.startswith(MOVE):
            return "valu", "move"
        if op.startswith("v_cndmask"):
            return "valu", "select"
        if op.startswith("v_cmp"):
            return "valu", "cmp"
        if op.startswith(TRANS):
            return "valu", "trans"
        if op.startswith(CVT):
            return "valu", "cvt"
        if op.startswith(ADDR):
            return "valu", "addr"
        return "valu", "fp"
    if op.startswith("ds_"):
        return "lds", None
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return ("vmem_st" if ("store" in op or "atomic" in op) else "vmem_ld"), None
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")) or "saveexec" in op:
        return "branch", None
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_wait")):
        return "wait", None
    if op.startswith("s_"):
        return "salu", None
    return "other", None


def compile_asm(src, defines, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out]  # = the Makefile's FLAGS
    cmd += ["-D" + d for d in defines] + extra
    subprocess.run(cmd, check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    if not names:
        return {}
    p = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True)
    dem = p.stdout.strip().split("\n")
    res = {}
    for n, d in zip(names, dem):
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*\)$", "", d)
        d = d.replace("scann::", "")
        res[n] = d
    return res


INSN = re.compile(r"^\t([a-z][a-z0-9_]+)(?:\s|$)")


def parse(asm_path):
    """-> {mangled kernel: {"ops": Counter(opcode), "meta": {...}}} for every .amdhsa_kernel of the file."""
    kernels = {}
    kernel_names = set()
    with open(asm_path) as f:
        lines = f.read().split("\n")
    for ln in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            kernel_names.add(m.group(1))
    cur = None
    for ln in lines:
        m = re.match(r"^([A-Za-z_][\w$.]*):", ln)
        if m and m.group(1) in kernel_names:
            cur = m.group(1)
            kernels[cur] = {"ops": collections.Counter(), "meta": {}}
            continue
        if cur:
            if ln.startswith(".Lfunc_end"):
                cur = None
                continue
            m = INSN.match(ln)
            if m and not m.group(1).startswith("."):
                kernels[cur]["ops"][m.group(1)] += 1
    for ln in lines:
        m = re.match(r"\s*\.set\s+(\S+)\.(num_vgpr|num_agpr|private_seg_size|numbered_sgpr),\s*(\S+)", ln)
        if m and m.group(1) in kernels:
            try:
                kernels[m.group(1)]["meta"][m.group(2)] = int(m.group(3))
            except ValueError:
                pass
    # LDS size lives in the kernel descriptor
    cur = None
    for ln in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            cur = m.group(1)
        m = re.match(r"\s*\.amdhsa_group_segment_fixed_size\s+(\d+)", ln)
        if m and cur in kernels:
            kernels[cur]["meta"]["lds"] = int(m.group(1))
    return kernels


def summarise(ops):
    tot = collections.Counter()
    sub = collections.Counter()
    for op, n in ops.items():
        c, s = classify(op)
        tot[c] += n
        if s:
            sub[s] += n
    return tot, sub


SUBS = ("fp", "cvt", "trans", "move", "select", "addr", "cmp", "xlane")
CLS = ("mfma", "valu", "lds", "vmem_ld", "vmem_st", "salu", "branch", "wait")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--src", default="scann_kernels.hip", help="file under scann--material_amd/csrc (or a path)")
    ap.add_argument("--asm", help="an already compiled .s file instead of --src")
    ap.add_argument("-k", "--kernel", action="append", help="substring of the demangled kernel name (repeatable); default: all")
    ap.add_argument("-v", "--verbose", action="store_true", help="opcode-by-opcode table of the selected kernels")
    ap.add_argument("-D", dest="defines", action="append", default=[])
    ap.add_argument("-X", dest="extra", action="append", default=[], help="extra hipcc argument")
    args = ap.parse_args()
    if args.asm:
        asm = args.asm
    else:
        src = args.src if os.path.isabs(args.src) else os.path.join(CSRC, args.src)
        asm = compile_asm(src, args.defines, args.extra)
    ks = parse(asm)
    names = demangle(sorted(ks))
    rows = []
    for mang, k in ks.items():
        name = names[mang]
        if args.kernel and not any(s in name for s in args.kernel):
            continue
        rows.append((name, k))
    rows.sort(key=lambda r: r[0])
    hdr = "%-46s %5s %5s %5s | %5s %5s %5s %5s %5s %5s %5s %5s | %4s %4s %4s %5s %5s %5s | %4s %5s %6s" % (
        ("kernel", "mfma", "valu", "=") + SUBS + ("lds", "vld", "vst", "salu", "brnch", "wait", "vgpr", "scr B", "lds B"))
    print(hdr)
    for name, k in rows:
        tot, sub = summarise(k["ops"])
        meta = k["meta"]
        print("%-46s %5d %5d %5s | %s | %4d %4d %4d %5d %5d %5d | %4d %5d %6d" % (
            name[:46], tot["mfma"], tot["valu"], "", " ".join("%5d" % sub[s] for s in SUBS), tot["lds"], tot["vmem_ld"], tot["vmem_st"],
            tot["salu"], tot["branch"], tot["wait"], meta.get("num_vgpr", -1), meta.get("private_seg_size", -1), meta.get("lds", -1)))
        if args.verbose:
            by = collections.defaultdict(list)
            for op, n in k["ops"].items():
                c, s = classify(op)
                by[(c, s)].append((n, op))
            for key in sorted(by, key=lambda x: (x[0], x[1] or "")):
                items = sorted(by[key], reverse=True)
                print("    %-14s %5d : %s" % ("/".join(x for x in key if x), sum(n for n, _ in items), "  ".join("%s %d" % (op, n) for n, op in items)))
    if not args.asm:
        os.unlink(asm)
    return 0


if __name__ == "__main__":
    sys.exit(main())
