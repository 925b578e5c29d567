cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_chunked_tracker.py tests/test_library_abi.py -m gpu -x -q > gpurun_out/r6_t1.txt 2>&1
tail -15 gpurun_out/r6_t1.txt
python tools/probe_tracker_split.py > gpurun_out/r6_split1.txt 2>&1; cat gpurun_out/r6_split1.txt | tail -4
python bench.py --steps 5 --warmup 2 --no-live-traffic --cpu-frames 0 > gpurun_out/r6_bench1.json 2> gpurun_out/r6_bench1.err; tail -c 3000 gpurun_out/r6_bench1.json
