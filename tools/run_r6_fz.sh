cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_chunked_tracker.py tests/test_gpu_corr.py -m gpu -x -q > gpurun_out/r6_t5.txt 2>&1
tail -6 gpurun_out/r6_t5.txt
python tools/bench_lookup_conv.py
python tools/probe_tracker_split.py 2>&1 | tail -3
