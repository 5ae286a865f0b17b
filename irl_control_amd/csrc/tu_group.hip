// Translation unit of the fp32 group path: osc_group_kernel_f32 (three Dual-UR5 shapes) and the give-up-list kernel.
#include "osc_generic.hpp"
#include "osc_group.hpp"
#include "launchers.hpp"

namespace irlosc {

// Give-up lists of a train (normally all empty): grid (16, nsteps), one wave per instance, grid-strided per list.
__global__ __launch_bounds__(64) void osc_generic_lists_kernel(const TrainStep* __restrict__ table) {
    extern __shared__ __align__(16) unsigned char smem_raw_l[];
    float* smem = reinterpret_cast<float*>(smem_raw_l);
    typedef const __attribute__((address_space(4))) TrainStep* ctab_t;
    const ctab_t ct = (ctab_t)table;
    const __attribute__((address_space(4))) S2Args& a = ct[blockIdx.y].prev;
    if (a.nfast <= 0) return;
    const int count = *a.workcount2;
    for (int it = blockIdx.x; it < count; it += gridDim.x) generic_instance<float>(ct[blockIdx.y].prev_p, a.worklist2[it], smem);
}

// The fused train launch: table[0..nsteps) on the device, total_blocks = sum of (riders + tiles) over the steps.
int launch_group_train(const TrainStep* dtable, int nsteps, int total_blocks, int k, int ndev, hipStream_t st) {
    if (total_blocks <= 0) return 0;
    const dim3 grid(total_blocks);
    if (k == 13 && ndev == 3) hipLaunchKernelGGL((osc_group_kernel_f32<4, 13, 3, 2>), grid, dim3(64), 0, st, dtable, nsteps);
    else if (k == 12 && ndev == 2) hipLaunchKernelGGL((osc_group_kernel_f32<4, 12, 2, 2>), grid, dim3(64), 0, st, dtable, nsteps);
    else if (k == 7 && ndev == 3) hipLaunchKernelGGL((osc_group_kernel_f32<4, 7, 3, 2>), grid, dim3(64), 0, st, dtable, nsteps);
    else return (int)hipErrorNotSupported;
    return (int)hipGetLastError();
}

// The generic kernel over the give-up lists of the steps whose stage 2 rode in `dtable`'s train.
int launch_giveup_lists(const TrainStep* dtable, int nsteps, int n, int k, int ndev, hipStream_t st) {
    if (nsteps <= 0) return 0;
    hipLaunchKernelGGL(osc_generic_lists_kernel, dim3(16, nsteps), dim3(64), generic_smem_bytes<float>(n, k, ndev), st, dtable);
    return (int)hipGetLastError();
}

}  // namespace irlosc
