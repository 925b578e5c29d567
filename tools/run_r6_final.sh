cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gputest.txt 2>&1; tail -3 gpurun_out/r06_gputest.txt
timeout 1500 python -m pytest tests -m gpu -q --conv-bf16x3 > gpurun_out/r06_gputest_conv_bf16x3.txt 2>&1; tail -3 gpurun_out/r06_gputest_conv_bf16x3.txt
bash tools/profile_r06.sh > gpurun_out/r06_profile.log 2>&1; tail -5 gpurun_out/r06_profile.log
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -2 gpurun_out/r06_bench_line.err
python tools/bench_pose_solve.py > gpurun_out/r06_pose_solve_ab.txt 2>&1
python tools/bench_lookup_conv.py > gpurun_out/r06_lookup_conv_ab.txt 2>&1
