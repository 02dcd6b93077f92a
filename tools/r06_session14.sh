#!/bin/bash
# WaveFrontSynchro with few frames: a CU's chroma modes posted to idle workgroups while at least N of them are idle (-DHEVCDL_CHROMA_ROOM=N variants) against the shipped build
rm -f gpurun_out/r06m_room.txt
for rep in 1 2; do
for l in libhevcdl_hip.so ab_room10.so ab_room30.so ab_room60.so; do
  echo "== $l rep $rep" >> gpurun_out/r06m_room.txt
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l timeout 200 python tools/time_rd.py 1 2 4 --wavefront 2>&1 | grep flags | cut -c1-70 >> gpurun_out/r06m_room.txt
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l timeout 200 python tools/time_rd.py 10 --size=1920x1080 --wavefront 2>&1 | grep flags | cut -c1-70 >> gpurun_out/r06m_room.txt
  HEVCDL_LIB=hevc-deep-learning-pipeline_amd/lib/$l timeout 200 python tools/time_rd.py 20 75 2>&1 | grep flags | cut -c1-70 >> gpurun_out/r06m_room.txt
done
done
cat gpurun_out/r06m_room.txt
