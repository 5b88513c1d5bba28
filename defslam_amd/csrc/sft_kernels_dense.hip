// Dense build of the SfT kernels: the same source compiled for four wavefronts per SIMD (128 VGPRs per wave) so that two
// 8-wavefront problems are resident per CU.  Everything in sft_kernels.hip lives in an anonymous namespace; only the two
// extern "C" entry points get their own names here.
#define SFT_WAVES_PER_EU 4
#define SFT_LAUNCH_NAME sft_lm_launch_dense
#define SFT_LDS_BYTES_NAME sft_lm_kernel_lds_bytes_dense
#define SFT_NO_ASSEMBLY_KERNEL   // one copy of the measurement kernel is enough
#include "sft_kernels.hip"
