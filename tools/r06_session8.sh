#!/bin/bash
# round 6: how far ahead should the row above be before a row becomes claimable?  (2 = what its first CTU needs; more = slack against stalls later in the row)
for lag in 2 3 4 6 8 12; do
  echo "== lag $lag"
  HEVCDL_WPP_LAG=$lag timeout 300 python tools/time_rd.py 75 150 300 600 --wavefront 2>&1 | grep flags
done > gpurun_out/r06g_wpp_lag.txt 2>&1
cat gpurun_out/r06g_wpp_lag.txt | cut -c1-150
