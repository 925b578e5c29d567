#!/bin/bash
# Where does the tail of k_pose_reduce spend its time?  Builds a probe variant of pose.hip ON THE GPU BOX (-DRPE_POSE_PROBE: 100 MHz clock
# stamps of row 0's tail), links it with the other objects, and runs a 16-row 640x512 solve through it.
set -e
cd /root/repo/robust-pose-estimator_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DRPE_POSE_PROBE -c pose.hip -o /tmp/pose_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librpe_probe.so /tmp/pose_probe.o $(ls build/*.o | grep -v build/pose.o)
cd /root/repo
RPE_HIP_LIBRARY=/tmp/librpe_probe.so python tools/probe_pose_tail.py
