python bench.py > gpurun_out/r4b_bench_line.json 2> gpurun_out/r4b_bench_err.txt; tail -c 600 gpurun_out/r4b_bench_err.txt
tools/profile_round.sh r04b > gpurun_out/r4b_profile_round.txt 2>&1
python tools/balanced_full_time.py > gpurun_out/r4b_balanced.txt 2>&1; tail -3 gpurun_out/r4b_balanced.txt
python tools/create_time.py > gpurun_out/r4b_create_time.txt 2>&1; tail -3 gpurun_out/r4b_create_time.txt
