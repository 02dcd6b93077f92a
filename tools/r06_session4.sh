#!/bin/bash
# round 6, fourth GPU session: how many waves of a workgroup should claim wavefront rows (the rest help them)?  75 / 150 / 600 frames of 2160p, eight- and ten-wave build
for fl in 2 4; do
for m in 2 3 4 5 6 8 10; do
  echo "== build flags $fl masters $m"
  HEVCDL_WPP_MASTERS=$m timeout 300 python tools/time_rd.py 75 150 600 --wavefront --flags=$fl 2>&1 | grep flags
done
done > gpurun_out/r06d_wpp_masters.txt 2>&1
for r in 1 2 3; do echo "== remote $r"; HEVCDL_WPP_REMOTE=$r timeout 300 python tools/time_rd.py 1 2 4 --wavefront 2>&1 | grep flags; HEVCDL_WPP_REMOTE=$r timeout 300 python tools/time_rd.py 10 --size=1920x1080 --wavefront 2>&1 | grep flags; done > gpurun_out/r06d_wpp_remote.txt 2>&1
cat gpurun_out/r06d_wpp_masters.txt gpurun_out/r06d_wpp_remote.txt | cut -c1-150
