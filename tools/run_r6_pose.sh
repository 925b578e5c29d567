cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_gpu_pose_backward.py tests/test_gpu_reference_golden.py tests/test_gpu_fullsize.py tests/test_gpu_chunked_tracker.py -m gpu -x -q > gpurun_out/r6_t3.txt 2>&1
tail -6 gpurun_out/r6_t3.txt
python bench.py --steps 10 --warmup 3 --no-live-traffic --cpu-frames 0 > gpurun_out/r6_bench3.json 2> gpurun_out/r6_bench3.err; tail -3 gpurun_out/r6_bench3.err
python - <<'P'
import json; d=json.load(open('gpurun_out/r6_bench3.json'))
for k in d:
    if k.startswith('tracker_') and 'config' not in k or k in ('value','ms_per_step','latency_batch1_ms','gn_ms_per_step'): print(k, d[k])
print(d['roofline']['frac'], d['roofline_pose_solve'], d['roofline_pose_solve_gn'])
P
