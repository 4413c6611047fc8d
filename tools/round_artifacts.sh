# the round's kept artifacts (GPU box, repo root): bench line, rocprofv3 kernel stats, the slow GPU tests, side measurements
python bench.py > gpurun_out/r5_bench_line.json 2> gpurun_out/r5_bench_err.txt; tail -c 600 gpurun_out/r5_bench_err.txt
tools/profile_round.sh r05 > gpurun_out/r5_profile_round.txt 2>&1
LT_RUN_SLOW=1 python -m pytest tests -q -m "gpu and slow" > gpurun_out/r5_slow_tests.txt 2>&1; tail -3 gpurun_out/r5_slow_tests.txt
python tools/balanced_full_time.py > gpurun_out/r5_balanced.txt 2>&1; tail -3 gpurun_out/r5_balanced.txt
