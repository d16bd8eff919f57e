# Round-6 measurement pass on the final code (run on the GPU box from the repo root: bash tools/prof_r6.sh): bench lines (default and the
# driver's arguments), rocprofv3 kernel-trace summaries at both launch shapes, every counter group of tools/pmc_r6.py plus FETCH_SIZE /
# WRITE_SIZE (separate passes) at both, one-batch latency, the other configurations.  --pmc is never combined with a tracing domain.
# gpurun brings back at most 64 MiB: the per-dispatch CSVs are reduced to per-kernel averages here (tools/pmc_report.py) and dropped.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python3 $R/bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err; echo "bench default rc $?"
python3 $R/bench.py --steps 20 --warmup 5 > $O/r06_bench_steps20.json 2> $O/r06_bench_steps20.err; echo "bench steps20 rc $?"
cut -c1-300 $O/r06_bench_steps20.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_g16 -- python3 $R/bench.py --no-extras --steps 400 > $O/prof_r06_g16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_g10 -- python3 $R/bench.py --no-extras --steps 20 --warmup 5 > $O/prof_r06_g10.log 2>&1
cp $O/prof_r06_g16/*/*kernel_stats.csv $O/r06_kernel_stats_g16.csv; cp $O/prof_r06_g10/*/*kernel_stats.csv $O/r06_kernel_stats_g10.csv
rm -rf $O/prof_r06_g16 $O/prof_r06_g10
head -8 $O/r06_kernel_stats_g16.csv; head -8 $O/r06_kernel_stats_g10.csv
cd $R
for tag in g10:20 g16:64; do t=${tag%%:*}; st=${tag##*:}; warm=$(( st < 32 ? st / 2 : 16 ))
  python3 tools/pmc_r6.py r06f $t 2>&1 | tee -a $O/r06f_pmc.log
  args="$R/bench.py --no-extras --steps $st --warmup $warm --min-time 0.3 --prewarm 0.3"
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r06f${t}_fetch -- python3 $args > $O/pmc_r06f${t}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_r06f${t}_write -- python3 $args > $O/pmc_r06f${t}_write.log 2>&1 )
  python3 tools/pmc_report.py r06f$t --traffic qm9_$t > /dev/null
  cp profiles/r06f${t}_counters.txt $O/
  rm -rf $O/pmc_r06f${t}_*
done
cp profiles/edge_kernel.json $O/edge_kernel.json
python3 tools/pmc_r6_table.py r06fg10 r06fg16 > $O/r06_memory_path.txt; grep -c "edge kernel" $O/r06_memory_path.txt
python3 tools/predict_latency.py > $O/r06_predict_latency.txt 2>&1; tail -6 $O/r06_predict_latency.txt
python3 tools/config_rates.py > $O/r06_config_rates.txt 2>&1; tail -5 $O/r06_config_rates.txt
du -sh $O
