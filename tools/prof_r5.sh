# Round-5 measurement pass (run on the GPU box from the repo root: bash tools/prof_r5.sh): the default and the driver-shape bench
# lines, rocprofv3 kernel-trace summaries at both launch shapes, FETCH_SIZE / WRITE_SIZE (separate passes) at both, SQ counters at
# 16 batches per launch.  --pmc is never combined with a tracing domain.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/r05_bench_steps20.json 2> $O/r05_bench_steps20.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_g16 -- python3 $R/bench.py --no-extras --steps 400 > $O/prof_r05_g16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_g10 -- python3 $R/bench.py --no-extras --steps 20 --warmup 5 > $O/prof_r05_g10.log 2>&1
for tag in g16:64 g10:20; do t=${tag%%:*}; st=${tag##*:}; warm=$(( st < 32 ? st / 2 : 16 ))
 args="$R/bench.py --no-extras --steps $st --warmup $warm --min-time 0.3 --prewarm 0.3"
 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r05${t}_fetch -- python3 $args > $O/pmc_r05${t}_fetch.log 2>&1
 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_r05${t}_write -- python3 $args > $O/pmc_r05${t}_write.log 2>&1
done
args="$R/bench.py --no-extras --steps 64 --warmup 16 --min-time 0.3 --prewarm 0.3"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r05g16_sq -- python3 $args > $O/pmc_r05g16_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_r05g16_sq2 -- python3 $args > $O/pmc_r05g16_sq2.log 2>&1

head -12 $O/prof_r05_g16/*/*kernel_stats.csv; head -12 $O/prof_r05_g10/*/*kernel_stats.csv; python3 $R/tools/pmc_report.py r05g16 --traffic qm9_g16 > /dev/null; python3 $R/tools/pmc_report.py r05g10 --traffic qm9_g10 > /dev/null; cp $R/profiles/r05g1*_counters.txt $R/profiles/edge_kernel.json $O/; cut -c1-400 $O/r05_bench_steps20.json
