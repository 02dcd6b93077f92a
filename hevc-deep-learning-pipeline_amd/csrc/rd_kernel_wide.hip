// The CTU-decision kernel with ten wavefronts per workgroup (8-bit samples): rd_kernel.hip compiled with HEVCDL_NW 10.
// Exports hevcdl_rd_frame_kernel_wide, hevcdl_rd_smem_bytes_wide, hevcdl_rd_scratch_bytes_wide, hevcdl_rd_waves_per_group_wide.
//
// Why a second build: a wave of this kernel spends most of its cycles waiting on its own dependent operations (LDS round trips, ordered fp64 sums), so a CU's
// throughput grows almost linearly with the waves resident on it (tools/micro_rd.py on the leaf routines of a TU coding: 8 -> 10 -> 12 waves per CU = +20 % -> +40 %
// codings per cycle).  Residency is bound by LDS (a wave's private block) and by registers (512 per SIMD lane: 2 waves at 256, 3 at 168).
// This build drops the look-ahead region (it only serves workgroups with ONE master: launches of few units, which keep the 8-wave kernel) and is held to 168
// registers by its launch bound.  The whole kernel gains less than its leaves, and only with two or more masters per workgroup (600 frames of 2160p on 256 CUs
// 6.24 -> 6.11 s, 1024 frames 10.33 -> 9.50 s, 2560 frames 21.47 -> 18.14 s; 300 frames 4.16 -> 4.59 s) and it moves three times the bytes through the L2s (spills,
// snapshots in HBM): launch_rd (hevcdl_api.hip) picks it from three units per workgroup on (four until round 5).
#define HEVCDL_NW 10
#undef HEVCDL_NPEND
#define HEVCDL_NPEND 2           // (an experiment with more pending passes in the eight-wave build leaves this one alone: its LDS is full)
#define HEVCDL_AHEAD 0
#define HEVCDL_RD_WIDE 1
#undef HEVCDL_KERNEL_PROF       // the in-kernel timers belong to the 8-wave build (no LDS to spare here)
#include "rd_kernel.hip"
