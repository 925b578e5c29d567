cd /root/repo
python tools/probe_tracker_split.py > gpurun_out/r6_split0.txt 2>&1
python tools/profile_tracker_host.py > gpurun_out/r6_hostprof0.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest0.txt 2>&1
tail -3 gpurun_out/r6_gputest0.txt; cat gpurun_out/r6_split0.txt
