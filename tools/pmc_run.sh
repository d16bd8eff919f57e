#!/bin/bash
# rocprofv3 counter passes for the bench workload (run on the GPU box from the repo root):
#   tools/pmc_run.sh <tag> [steps]  -> gpurun_out/pmc_<tag>_{sq,sq2,fetch,write}/ + kernel stats gpurun_out/prof_<tag>/
# Separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with a tracing domain.
set -u
tag=${1:-r02}
steps=${2:-64}     # bench --steps of the counter passes: 64 -> four 16-batch groups; 20 -> the driver's 10 + 10
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
warm=$(( steps < 32 ? steps / 2 : 16 ))   # warm-up launches of the same group size as the measured ones
args="$root/bench.py --no-extras --steps $steps --warmup $warm --min-time 0.3 --prewarm 0.3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 $root/bench.py --no-extras --steps 400 > $out/prof_$tag.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_${tag}_sq -- python3 $args > $out/pmc_${tag}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_${tag}_sq2 -- python3 $args > $out/pmc_${tag}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${tag}_fetch -- python3 $args > $out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${tag}_write -- python3 $args > $out/pmc_${tag}_write.log 2>&1
ls $out/prof_$tag/*/ $out/pmc_${tag}_sq/*/ | head -20
