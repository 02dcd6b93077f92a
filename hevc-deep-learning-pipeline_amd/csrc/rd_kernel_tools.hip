// The CTU-decision kernel with the cfg's tool switches read at run time (8-bit samples, eight wavefronts): rd_kernel.hip compiled with HEVCDL_TOOLS_RT 1.
// Exports hevcdl_rd_frame_kernel_tools, hevcdl_rd_smem_bytes_tools, hevcdl_rd_scratch_bytes_tools, hevcdl_rd_waves_per_group_tools.
//
// Why a build of its own: the timed configurations run the reference cfg's tools (RDOQ, RDOQTS, TransformSkip, TransformSkipFast, SignHideFlag, StrongIntraSmoothing,
// FastUDIUseMPMEnabled all 1); with the switches as compile-time constants in rd_kernel.hip / rd_kernel_wide.hip every test on them folds away and the quantiser without
// RDOQ is not part of their code.  launch_rd (hevcdl_api.hip) takes this build for a context whose hevcdl_config.tools differs (tests/golden/rd_k*.npz pin it).
#define HEVCDL_RD_TOOLS 1
#define HEVCDL_TOOLS_RT 1
#undef HEVCDL_KERNEL_PROF       // the in-kernel timers belong to the reference-tools build
#include "rd_kernel.hip"
