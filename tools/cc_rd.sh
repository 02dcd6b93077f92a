#!/bin/bash
# compile csrc/rd_kernel.hip alone (gfx950) and print the kernel's resource usage: tools/cc_rd.sh [extra hipcc flags]
C=/root/repo/hevc-deep-learning-pipeline_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -amdgpu-spill-vgpr-to-agpr=0 \
  -I/root/repo/include -I$C "$@" -c $C/rd_kernel.hip -o /tmp/rd_kernel.o -save-temps=obj 2>&1 | grep -E "error|static assertion" -A3 | head -40
grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count|vgpr_spill_count):" /tmp/rd_kernel-hip-amdgcn-amd-amdhsa-gfx950.s
