// nmi_kernels_stamped.hip -- nmi_grid_kernel once more, as nmi_grid_kernel_stamped + launch_grid_stamped: the same code with
// wall-clock stamps at the phase boundaries of every workgroup's first candidate (NMI_OPT_STAMPS; tools/grid_stamps.py).
// A translation unit of its own so that the product's kernel is not touched by the instrumentation.
#define NMI_GRID_KERNEL_STAMPED 1
#include "nmi_kernels.hip"
