#!/bin/bash
# round 6: the scheduling thresholds of the decision kernel once more on the final leaves (variants built with -D..., tools/time_rd.py 1 / 75 / 600 frames, two rounds)
rm -f gpurun_out/r06l_knobs.txt
for rep in 1 2; do
for l in libhevcdl_hip.so ab_carry3.so ab_carry1.so ab_pre2.so ab_pre4.so ab_hop2.so ab_hop8.so ab_ahead2.so ab_ahead3.so ab_sleep4.so ab_sleep16.so ab_fg3.so ab_fg1.so ab_slice2.so ab_slice4.so; do
  echo "== $l rep $rep" >> gpurun_out/r06l_knobs.txt
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l timeout 200 python tools/time_rd.py 1 75 600 2>&1 | grep flags | cut -c1-60 >> gpurun_out/r06l_knobs.txt
done
done
cat gpurun_out/r06l_knobs.txt
