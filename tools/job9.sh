cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PROF_GLUE=1 python tools/phase_profile.py 3840 2160 600 > gpurun_out/prof5g_f600.txt 2>&1
grep -E "leaf|split_bits|put back|saved|before the search|task:|KERNEL|chain owner|code_tu_block|intra_bits|IDLE|MASTER" gpurun_out/prof5g_f600.txt | cut -c1-130
