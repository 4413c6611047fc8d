# the round's kept artifacts (GPU box, repo root): bench line, rocprofv3 kernel stats, the slow GPU tests, side measurements
python bench.py > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench_err.txt; tail -c 600 gpurun_out/r6_bench_err.txt
tools/profile_round.sh r06 > gpurun_out/r6_profile_round.txt 2>&1
LT_RUN_SLOW=1 python -m pytest tests -q -m "gpu and slow" > gpurun_out/r6_slow_tests.txt 2>&1; tail -3 gpurun_out/r6_slow_tests.txt
python tools/shard_time.py 21 delta,sparse > gpurun_out/r6_shard_time.txt 2>&1; tail -2 gpurun_out/r6_shard_time.txt
(python tools/shard_step_time.py delta 500; python tools/shard_step_time.py delta 2000) > gpurun_out/r6_shard_step.txt 2>&1; tail -8 gpurun_out/r6_shard_step.txt
python tools/stageb_lab2.py delta > gpurun_out/r6_stageb_lab2.txt 2>&1; tail -9 gpurun_out/r6_stageb_lab2.txt
python tools/hub_noise.py > gpurun_out/r6_hub_noise.txt 2>&1; tail -3 gpurun_out/r6_hub_noise.txt
python tools/balanced_full_time.py > gpurun_out/r6_balanced.txt 2>&1; tail -3 gpurun_out/r6_balanced.txt
