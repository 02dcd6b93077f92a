// The CTU-decision kernel for 10-bit samples (InternalBitDepth 10, uint16 planes): rd_kernel.hip compiled with HEVCDL_BD 10.
// Exports hevcdl_rd_frame_kernel_bd10, hevcdl_rd_smem_bytes_bd10, hevcdl_rd_scratch_bytes_bd10.
#define HEVCDL_BD 10
#undef HEVCDL_KERNEL_PROF       // the in-kernel timers are an 8-bit affair (no LDS to spare here)
#include "rd_kernel.hip"
