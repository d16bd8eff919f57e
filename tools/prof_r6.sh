# Round-6 measurement pass on the final code (run on the GPU box from the repo root: bash tools/prof_r6.sh [part]): bench lines (default
# and the driver's arguments), rocprofv3 kernel-trace summaries at both launch shapes, FETCH_SIZE / WRITE_SIZE (separate passes) at both,
# every counter group of tools/pmc_r6.py at both, one-batch latency, the other configurations.  --pmc is never combined with a tracing domain.
set -u
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
part=${1:-all}
if [ $part = all ] || [ $part = bench ]; then
python3 $R/bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err; echo "bench default rc $?"
python3 $R/bench.py --steps 20 --warmup 5 > $O/r06_bench_steps20.json 2> $O/r06_bench_steps20.err; echo "bench steps20 rc $?"
cut -c1-300 $O/r06_bench_steps20.json
fi
if [ $part = all ] || [ $part = trace ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_g16 -- python3 $R/bench.py --no-extras --steps 400 > $O/prof_r06_g16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_g10 -- python3 $R/bench.py --no-extras --steps 20 --warmup 5 > $O/prof_r06_g10.log 2>&1
for tag in g16:64 g10:20; do t=${tag%%:*}; st=${tag##*:}; warm=$(( st < 32 ? st / 2 : 16 ))
 args="$R/bench.py --no-extras --steps $st --warmup $warm --min-time 0.3 --prewarm 0.3"
 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r06f${t}_fetch -- python3 $args > $O/pmc_r06f${t}_fetch.log 2>&1
 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_r06f${t}_write -- python3 $args > $O/pmc_r06f${t}_write.log 2>&1
done
head -12 $O/prof_r06_g16/*/*kernel_stats.csv; head -12 $O/prof_r06_g10/*/*kernel_stats.csv
cd $R
fi
if [ $part = all ] || [ $part = pmc ]; then
cd $R && python3 tools/pmc_r6.py r06f 2>&1 | tee $O/r06f_pmc.log
fi
if [ $part = all ] || [ $part = legs ]; then
cd $R
python3 tools/predict_latency.py > $O/r06_predict_latency.txt 2>&1; tail -6 $O/r06_predict_latency.txt
python3 tools/config_rates.py > $O/r06_config_rates.txt 2>&1; tail -8 $O/r06_config_rates.txt
fi
