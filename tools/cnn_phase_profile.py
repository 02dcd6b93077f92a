#!/usr/bin/env python3
"""In-kernel phase profile of the CNN kernel (build with -DHEVCDL_CNN_PROF into lib/libhevcdl_hip_cnnprof.so)."""
import os, subprocess, sys
code = """
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
yuv=ref_tools.synth_yuv(1920,1080,4,seed=1)
enc=hevcdl_amd.Encoder(1920,1080,32,max_frames=4); lab,lg=enc.predict_depth(yuv,want_logits=True)
names=['prologue','conv64 (2 halves, tiles incl.)','conv1 tile x4','conv1 x4','conv2 x4','conv3 x4','-','epilogue']   # the fully connected head is fc_kernel.hip
v=lg.reshape(-1)[:8]; tot=v.sum()
for n,c in zip(names,v): print('%-18s %10.0f cycles %5.1f%%'%(n,c,100*c/tot))
print('total %.0f cycles (all CUs busy: 2040 CTUs per frame)'%tot)
enc.close()
"""
out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HEVCDL_LIB="/root/repo/hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_cnnprof.so"), capture_output=True, text=True)
print(out.stdout, out.stderr[-400:])
