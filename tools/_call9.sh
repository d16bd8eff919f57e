set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 400 python3 tests/manual/parity_s134k.py > $O/r06_parity_s134k.txt 2>&1; echo "s134k rc $?"; tail -6 $O/r06_parity_s134k.txt
timeout -k 10 300 python3 tests/manual/fuzz_parity.py 240 > $O/r06_fuzz_parity.txt 2>&1; echo "fuzz parity rc $?"; tail -6 $O/r06_fuzz_parity.txt
timeout -k 10 460 python3 tests/manual/fuzz_grads.py 420 census=6 > $O/r06_fuzz_census_long.txt 2>&1; echo "fuzz grads rc $?"; tail -7 $O/r06_fuzz_census_long.txt
