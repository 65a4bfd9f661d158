// nmi_kernels_rows.hip -- nmi_grid_kernel once more, as nmi_grid_kernel_rows + launch_grid_rows: the form for frames whose rows
// are not whole aligned 16-byte chunks (width % 16 != 0 -- KITTI's 1241 x 376 -- or stacks that are not 16-byte aligned).  Same
// kernel; the histogram phase addresses chunks per row with unaligned 16-byte loads and adds each row's last width % 16 pixels
// one by one (histogram_phase<..., ROWS = true> in nmi_kernels.hip).  A translation unit of its own so that the product's main
// kernel -- at its register cap -- is not touched.
#define NMI_GRID_KERNEL_ROWS 1
#include "nmi_kernels.hip"
