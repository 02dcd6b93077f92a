"""Timeline of workgroup 0 over four CTUs of a one-frame launch (-DHEVCDL_KERNEL_DEBUG -DHEVCDL_TIMELINE build of the library):
    HEVCDL_LIB=.../libhevcdl_hip_tl.so python tools/timeline.py [frames]  -> events sorted by time, per wave: clock (kilocycles from the first event), wave, event, argument"""
import os, subprocess, sys
code = """
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W,H=3840,2160
F=%d
yuv=np.concatenate([ref_tools.synth_yuv(W,H,min(F,4),seed=1)]*((F+3)//4))[:F]
enc=hevcdl_amd.Encoder(W,H,32,max_frames=max(F,256)); lab=enc.predict_depth(yuv); enc.compress_frames(yuv,lab); enc.close()
""" % (int(sys.argv[1]) if len(sys.argv) > 1 else 1)
out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
ev = [tuple(int(v) for v in l.split()[1:3]) for l in out.stdout.splitlines() if l.startswith('DBGV')]
names = {1: 'luma: start', 2: 'luma: P1 region (arg: claimed ahead)', 3: 'luma: P1 done', 4: 'luma: end', 5: 'chroma: start', 6: 'chroma: posted', 7: 'ahead: SATD done', 8: 'ahead: opened',
         9: 'chroma: answers in', 10: 'chroma: end', 11: 'CU syntax done', 12: 'CU logged',
         13: 'CTU: start', 14: 'CTU: walk done (arg: passes pending)', 15: 'CTU: passes joined', 16: 'CTU: state advanced'}
kinds = {1: 'P1', 2: 'chroma', 3: 'split', 4: 'P2', 5: 'RMD', 6: 'AHEAD'}
if not ev:
    print(out.stderr[-2000:]); sys.exit(1)
ev.sort(key=lambda e: e[0])
t0 = ev[0][0]
for t, w in ev:
    wave, e, arg = w >> 24, (w >> 16) & 255, w & 0xffff
    if 30 <= e < 40: nm = {30: 'luma: winner known', 31: 'luma: next SATD slices open', 32: 'luma: arrays in', 33: 'luma: winner copied', 34: 'luma: before P2 post', 35: 'luma: P2 (+chroma) posted'}.get(e, str(e))
    elif e >= 40: nm = 'task end   %s' % kinds.get(e - 40, e - 40)
    elif e >= 20: nm = 'task start %s' % kinds.get(e - 20, e - 20)
    else: nm = names.get(e, str(e))
    print("%9.1f  w%d  %s%s  %d" % (((t - t0) & 0xffffffff) / 1000.0, wave, '    ' * (1 if wave else 0), nm, arg))
