import sys, os, subprocess
names=["build_refs","filter_refs","rmd_satd","predict_block","rdoq:cg-prologue","rdoq:cg-walk","rdoq","dequant","rdoq:cg-epilogue","code_tu_block(total)","intra_bits_qt(total)","code_coeff_lane0","cabac_copy","enc_cu_syntax","KERNEL","set_result_cu","est_luma(total)","est_chroma(total)","rdoq:phaseA","rdoq:tail","rdoq:CGloop","rdoq:lastpos","rdoq:sbh","IDLE helper: no task","tu:refs+pred","tu:org+residual","tu:fwd","tu:rdoq(mark)","tu:store+dequant+inv","tu:recon+sse","load_tu_coef","MASTER time (process_unit)","IDLE master: tail of own region","IDLE region_wait (P2 join, carry)","rdoq:serial sums (per group)","rdoq:serial group logic (per group)","luma:pass1 region (wall)","chain owner: wait for its split tasks","restarts (calls) / CUs thrown away (kcycles)","chroma region (wall)","task: first-pass candidate","task: chroma mode","task: split of a child","task: deferred second pass","P2 join: no ticket region free","CTU: compress_cu (master)","CTU: encode_cu_tree","CTU: init + record flush","in tasks: code_tu_block","in tasks: intra_bits_qt","P1 task: prologue","P1 task: recur_luma","P1 task: epilogue","helper: import_owner","P2 join: end of the CTU (behind the encode)","P2 join: found finished at a CU start","TU 4x4 (total)","TU 8x8 (total)","TU 16x16 (total)","TU 32x32 (total)","rdoq 4x4","rdoq 8x8","rdoq 16x16","rdoq 32x32"]
if os.environ.get("PROF_GLUE"):      # library built with -DHEVCDL_KERNEL_PROF -DHEVCDL_PROF_N=64 -DHEVCDL_PROF_GLUE (csrc/rd_kernel.hip): the RDOQ phase accumulators carry these instead
    for i, n in ((19, "leaf (TU coding + bit count) in first-pass candidates"), (20, "leaf in chroma modes"), (21, "leaf in splits of a child"), (22, "leaf in the second pass's own chain"),
                 (34, "split_bits"), (35, "recur_luma: unsplit TU put back"), (4, "spec_children: own state saved / restored"), (5, "run_task: before the search"), (8, "(unused)"), (18, "(unused)")):
        names[i] = n
code="""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W,H=%s,%s
yuv=ref_tools.synth_yuv(W,H,1,seed=1)
F=int(sys.argv[3]) if len(sys.argv)>3 else 1
yuv=np.concatenate([ref_tools.synth_yuv(W,H,min(F,4),seed=1)]*((F+3)//4))[:F]
import os
enc=hevcdl_amd.Encoder(W,H,32,max_frames=F,wavefront=bool(os.environ.get('PROF_WAVEFRONT'))); lab=enc.predict_depth(yuv); enc.compress_frames(yuv,lab); enc.close()
"""%(sys.argv[1],sys.argv[2])
code=code.replace("sys.argv[3]", repr(sys.argv[3]) if len(sys.argv)>3 else "'1'").replace("len(sys.argv)>3","True")
out=subprocess.run([sys.executable,'-c',code],env=dict(os.environ,HEVCDL_LIB=os.environ.get('PROF_LIB','/root/repo/hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_prof.so')),capture_output=True,text=True)
rows=[l.split()[1:] for l in out.stdout.splitlines() if l.startswith('DBGV')]
tot=int(rows[14][0]) if len(rows)>14 else 1
for i,r in enumerate(rows[:len(names)]):
    print("%-24s kcycles %10d  calls %8d  %5.1f%%  cyc/call %d"%(names[i],int(r[0]),int(r[1]),100.0*int(r[0])/tot, 1024*int(r[0])//max(1,int(r[1]))))
print(out.stderr[-500:])
