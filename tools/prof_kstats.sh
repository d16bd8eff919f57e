# rocprofv3 kernel-trace summaries of the forward bench at the two launch shapes the bench line cites (16-batch groups; the driver's
# --steps 20 --warmup 5 = 10 + 10):  bash tools/prof_kstats.sh <tag>   ->  gpurun_out/prof_<tag>_g16 | _g10
set -u
tag=${1:-r3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_g16 -- python3 $R/bench.py --no-extras --steps 400 > $O/prof_${tag}_g16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_g10 -- python3 $R/bench.py --no-extras --steps 20 --warmup 5 > $O/prof_${tag}_g10.log 2>&1
head -8 $O/prof_${tag}_g16/*/*kernel_stats.csv; head -8 $O/prof_${tag}_g10/*/*kernel_stats.csv
