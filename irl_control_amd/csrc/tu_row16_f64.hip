// Translation unit of the fp64-arithmetic row16 path on double records: the three Dual-UR5 shapes and the give-up pass.
#include "osc_generic.hpp"
#include "osc_row16.hpp"
#include "topo_dual_ur5.hpp"     // the fused path exists for the compiled tree shape (irlosc_set_model checks the model against it)
#include "launchers.hpp"

namespace irlosc {

// nsteps steps of equal batch size B (the steps of one train), blockIdx.y = step
// tree: every record of the train carries the zero pattern of the compiled Dual-UR5 tree (irlosc.hip keeps that verdict per
// slot) -- the factorisation then runs in the tree-structured form on the dense records.
template <typename TIN>
int launch_row16(const Row16Train<TIN>& tr, int nsteps, bool tree, hipStream_t st) {
    const KParams<TIN>& p = tr.p[0];
    if (p.B <= 0 || nsteps <= 0) return 0;
    const dim3 grid((p.B + 3) / 4, nsteps);
    // part 1 of the task signal as a pass of its own (osc_task_rows_dense_kernel; no rows buffer: computed in the kernel)
    if (tr.x[0].trows) {
        hipLaunchKernelGGL((osc_task_rows_dense_kernel<TIN>), dim3((p.B + 63) / 64, nsteps), dim3(64 * p.ndev), 0, st, tr);
        const int rc = (int)hipGetLastError();      // (not left to the sticky last-error: the main launch below would be queued behind a failed pass)
        if (rc) return rc;
    }
    if (p.padded)      // every other n = 25 layout: the KMAX-padded variants (tu_row16_pad_impl.hpp)
        return tree ? launch_row16_pad_tree<TIN>(tr, nsteps, st) : launch_row16_pad_dense<TIN>(tr, nsteps, st);
    if (tree) {
        if (p.k == 13 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<13, 3, TIN, 25, false, TopoDualUr5>), grid, dim3(64), 0, st, tr);
        else if (p.k == 12 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<12, 2, TIN, 25, false, TopoDualUr5>), grid, dim3(64), 0, st, tr);
        else if (p.k == 7 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<7, 3, TIN, 25, false, TopoDualUr5>), grid, dim3(64), 0, st, tr);
        else if (p.k == 6 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<6, 2, TIN, 25, false, TopoDualUr5>), grid, dim3(64), 0, st, tr);
        else return (int)hipErrorNotSupported;
        return (int)hipGetLastError();
    }
    if (p.k == 13 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<13, 3, TIN, 25>), grid, dim3(64), 0, st, tr);
    else if (p.k == 12 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<12, 2, TIN, 25>), grid, dim3(64), 0, st, tr);
    else if (p.k == 7 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<7, 3, TIN, 25>), grid, dim3(64), 0, st, tr);
    else if (p.k == 6 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<6, 2, TIN, 25>), grid, dim3(64), 0, st, tr);
    else return (int)hipErrorNotSupported;
    return (int)hipGetLastError();
}

// The same train on the fused path: operands from the compact exchange buffers (tr.x[i].side / qvel / tables), the
// factorisation in the tree-structured form of the compiled Dual-UR5 shape.  Blocks of FOUR waves (256 threads): block x takes
// robots 16 (x % 4) .. of walk wave x / 4, i.e. one 128-byte line of every entry of that wave's exchange block.
template <typename TIN>
int launch_row16_fromq(const Row16Train<TIN>& tr, int nsteps, hipStream_t st, int parts) {
    const KParams<TIN>& p = tr.p[0];
    if (p.B <= 0 || nsteps <= 0) return 0;
    const int waves = (p.B + 63) / 64;
    const dim3 grid(waves * 4, nsteps), tgrid(waves, nsteps);      // the task pass first: one lane per robot, block = walk wave
    if (p.padded) return launch_row16_pad_fromq<TIN>(tr, nsteps, st, parts);
    if (!(parts & 1)) {}
    else if (p.k == 13 && p.ndev == 3) hipLaunchKernelGGL((osc_task_rows_fromq_kernel<13, 3, TIN, TopoDualUr5>), tgrid, dim3(64 * 3), 0, st, tr);
    else if (p.k == 12 && p.ndev == 2) hipLaunchKernelGGL((osc_task_rows_fromq_kernel<12, 2, TIN, TopoDualUr5>), tgrid, dim3(64 * 2), 0, st, tr);
    else if (p.k == 7 && p.ndev == 3) hipLaunchKernelGGL((osc_task_rows_fromq_kernel<7, 3, TIN, TopoDualUr5>), tgrid, dim3(64 * 3), 0, st, tr);
    else if (p.k == 6 && p.ndev == 2) hipLaunchKernelGGL((osc_task_rows_fromq_kernel<6, 2, TIN, TopoDualUr5>), tgrid, dim3(64 * 2), 0, st, tr);
    else return (int)hipErrorNotSupported;
    if (!(parts & 2)) return (int)hipGetLastError();
    if (p.k == 13 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<13, 3, TIN, 25, true, TopoDualUr5>), grid, dim3(256), 0, st, tr);
    else if (p.k == 12 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<12, 2, TIN, 25, true, TopoDualUr5>), grid, dim3(256), 0, st, tr);
    else if (p.k == 7 && p.ndev == 3) hipLaunchKernelGGL((osc_row16_kernel<7, 3, TIN, 25, true, TopoDualUr5>), grid, dim3(256), 0, st, tr);
    else if (p.k == 6 && p.ndev == 2) hipLaunchKernelGGL((osc_row16_kernel<6, 2, TIN, 25, true, TopoDualUr5>), grid, dim3(256), 0, st, tr);
    else return (int)hipErrorNotSupported;
    return (int)hipGetLastError();
}


// The generic kernel (Jacobi, fp64 arithmetic) over the give-up lists of a train; zeroes the counters `reset` points at.
template <typename TIN>
int launch_row16_worklist(const Row16Train<TIN>& tr, int nsteps, int32_t* reset, hipStream_t st) {
    const KParams<TIN>& p = tr.p[0];
    hipLaunchKernelGGL((osc_generic_worklist_kernel<double, TIN>), dim3(64, nsteps), dim3(64),
                       generic_smem_bytes<double>(p.n, p.k, p.ndev), st, tr, reset);
    return (int)hipGetLastError();
}

template int launch_row16<double>(const Row16Train<double>&, int, bool, hipStream_t);
template int launch_row16_fromq<double>(const Row16Train<double>&, int, hipStream_t, int);
template int launch_row16_worklist<double>(const Row16Train<double>&, int, int32_t*, hipStream_t);

void row16_tree_masks(uint32_t mrow[32], uint32_t* jcols) { r16::tree_structure_masks<TopoDualUr5>(mrow, jcols); }

}  // namespace irlosc
