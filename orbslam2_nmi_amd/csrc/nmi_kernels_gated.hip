// nmi_kernels_gated.hip -- nmi_grid_kernel once more, as nmi_grid_kernel_gated + launch_grid_gated: the fallback launch
// behind the few-levels kernels (nmi_fewlevels_kernel.hip).  See the comment above the kernel in nmi_kernels.hip.
#define NMI_GRID_KERNEL_GATED 1
#include "nmi_kernels.hip"
