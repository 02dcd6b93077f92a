import sys, os, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, ref_tools
import __graft_entry__ as g; g.build_oracle()
W,H,qp=64,64,32
kind = int(sys.argv[1]) if len(sys.argv)>1 else 3
yuv=ref_tools.synth_yuv(W,H,1,seed=1); lab=ref_tools.make_labels(W,H,1,kind,seed=2)
ref_tools.run_oracle(yuv,W,H,qp,lab,trace_path='/tmp/otrace.txt')
o=[l.strip() for l in open('/tmp/otrace.txt')]
code="""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
yuv=ref_tools.synth_yuv(%d,%d,1,seed=1); lab=ref_tools.make_labels(%d,%d,1,%d,seed=2)
enc=hevcdl_amd.Encoder(%d,%d,%d,max_frames=1); enc.compress_frames(yuv,lab); enc.close()
"""%(W,H,W,H,kind,W,H,qp)
out=subprocess.run([sys.executable,'-c',code],env=dict(os.environ,HEVCDL_DEBUG='1'),capture_output=True,text=True).stdout
gt=[l[2:].strip() for l in out.splitlines() if l.startswith('T ')]
print('\n'.join([l for l in out.splitlines() if l.startswith('PU')][:12]))
o=[l for l in o if not l.endswith(' 536870911') and int(l.split()[1])<500000000]; gt=[l for l in gt if not l.endswith(' 536870911') and int(l.split()[1])<500000000]
print("oracle trace",len(o),"gpu trace",len(gt))
for i,(a,b) in enumerate(zip(o,gt)):
    if a!=b:
        print("first diff at",i); 
        for j in range(max(0,i-64),min(len(o),i+4)): print(j,"oracle",o[j],"gpu",gt[j] if j<len(gt) else None)
        break
else: print("traces equal over common prefix")
