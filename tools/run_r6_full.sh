cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_full.txt 2>&1
tail -4 gpurun_out/r6_gputest_full.txt
python bench.py --steps 10 --warmup 3 --no-live-traffic > gpurun_out/r6_bench2.json 2> gpurun_out/r6_bench2.err; tail -3 gpurun_out/r6_bench2.err
