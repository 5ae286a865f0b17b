// The KMAX-padded row16 kernels (osc_row16.hpp, PAD): every n = 25 layout without an instantiation of its own -- single arm, arm +
// base, re-masked arms (osc.py:134-138, examples/ps_move_example.py:137-150) -- runs on the smallest tier KMAX in {4, 7, 10, 13, 16}
// that holds its k; k and ndev are kernel arguments.  One translation unit per (record type, form) so that the library still builds
// in parallel: each tu_row16_pad_*.hip defines IRLOSC_PAD_TIN and ONE of IRLOSC_PAD_DENSE / IRLOSC_PAD_TREE / IRLOSC_PAD_FROMQ.
#pragma once
#include "osc_generic.hpp"
#include "osc_row16.hpp"
#include "topo_dual_ur5.hpp"
#include "launchers.hpp"

namespace irlosc {

template <int KMAX, typename TIN, bool FROMQ, class TOPO>
static void pad_launch(const Row16Train<TIN>& tr, const dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((osc_row16_kernel<KMAX, IRLOSC_MAX_DEV, TIN, 25, FROMQ, TOPO, true>), grid, dim3(FROMQ ? 256 : 64), 0, st, tr);
}
template <typename TIN, bool FROMQ, class TOPO>
static int pad_dispatch(const Row16Train<TIN>& tr, const dim3 grid, hipStream_t st) {
    switch (row16_pad_tier(tr.p[0].k)) {
        case 4: pad_launch<4, TIN, FROMQ, TOPO>(tr, grid, st); break;
        case 7: pad_launch<7, TIN, FROMQ, TOPO>(tr, grid, st); break;
        case 10: pad_launch<10, TIN, FROMQ, TOPO>(tr, grid, st); break;
        case 13: pad_launch<13, TIN, FROMQ, TOPO>(tr, grid, st); break;
        case 16: pad_launch<16, TIN, FROMQ, TOPO>(tr, grid, st); break;
        default: return (int)hipErrorNotSupported;
    }
    return (int)hipGetLastError();
}

#if defined(IRLOSC_PAD_DENSE)
template <>
int launch_row16_pad_dense<IRLOSC_PAD_TIN>(const Row16Train<IRLOSC_PAD_TIN>& tr, int nsteps, hipStream_t st) {
    return pad_dispatch<IRLOSC_PAD_TIN, false, void>(tr, dim3((tr.p[0].B + 3) / 4, nsteps), st);
}
#elif defined(IRLOSC_PAD_TREE)
template <>
int launch_row16_pad_tree<IRLOSC_PAD_TIN>(const Row16Train<IRLOSC_PAD_TIN>& tr, int nsteps, hipStream_t st) {
    return pad_dispatch<IRLOSC_PAD_TIN, false, TopoDualUr5>(tr, dim3((tr.p[0].B + 3) / 4, nsteps), st);
}
#elif defined(IRLOSC_PAD_FROMQ)
// the fused path: task pass (one lane per (robot, device); the block has 64 x ndev threads), then the OSC kernel on the tile
template <>
int launch_row16_pad_fromq<IRLOSC_PAD_TIN>(const Row16Train<IRLOSC_PAD_TIN>& tr, int nsteps, hipStream_t st, int parts) {
    const KParams<IRLOSC_PAD_TIN>& p = tr.p[0];
    const int waves = (p.B + 63) / 64;
    if (parts & 1)
        hipLaunchKernelGGL((osc_task_rows_fromq_kernel<16, IRLOSC_MAX_DEV, IRLOSC_PAD_TIN, TopoDualUr5, true>), dim3(waves, nsteps),
                           dim3(64 * p.ndev), 0, st, tr);
    if (!(parts & 2)) return (int)hipGetLastError();
    return pad_dispatch<IRLOSC_PAD_TIN, true, TopoDualUr5>(tr, dim3(waves * 4, nsteps), st);
}
#else
#error "define IRLOSC_PAD_DENSE, IRLOSC_PAD_TREE or IRLOSC_PAD_FROMQ"
#endif

}  // namespace irlosc
