cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_chunked_tracker.py -m gpu -x -q > gpurun_out/r6_t2.txt 2>&1
tail -25 gpurun_out/r6_t2.txt
python tools/probe_tracker_split.py > gpurun_out/r6_split2.txt 2>&1; tail -4 gpurun_out/r6_split2.txt
python tools/profile_tracker_host.py > gpurun_out/r6_hostprof2.txt 2>&1
