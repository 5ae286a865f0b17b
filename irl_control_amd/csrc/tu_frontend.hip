// Translation unit of the wave-per-instance rigid-body front end (any tree of hinges), float and double records.
#include <cstring>

#include "osc_frontend.hpp"
#include "launchers.hpp"

namespace irlosc {

template <typename TOUT>
int launch_frontend_generic(const FeModel* dmodel, const double* qpos, const double* qvel, const FeOut<TOUT>& out, int B, size_t smem,
                            hipStream_t st) {
    if (B <= 0) return 0;
    FeGenericArgs<TOUT> a;
    memset(&a, 0, sizeof a);
    a.qpos[0] = qpos; a.qvel[0] = qvel; a.out[0] = out; a.B = B;
    hipLaunchKernelGGL(osc_frontend_kernel<TOUT>, dim3(B < (1 << 20) ? B : (1 << 20)), dim3(64), smem, st, dmodel, a);
    return (int)hipGetLastError();
}
template int launch_frontend_generic<float>(const FeModel*, const double*, const double*, const FeOut<float>&, int, size_t, hipStream_t);
template int launch_frontend_generic<double>(const FeModel*, const double*, const double*, const FeOut<double>&, int, size_t, hipStream_t);

// The same kernel over the device-side worklists of a train's steps (the give-up pass of the fused path): grid (64, nsteps).
template <typename TOUT>
int launch_frontend_generic_lists(const FeModel* dmodel, const FeGenericArgs<TOUT>& a, int nsteps, size_t smem, hipStream_t st) {
    if (a.B <= 0 || nsteps <= 0) return 0;
    hipLaunchKernelGGL(osc_frontend_kernel<TOUT>, dim3(64, nsteps), dim3(64), smem, st, dmodel, a);
    return (int)hipGetLastError();
}
template int launch_frontend_generic_lists<float>(const FeModel*, const FeGenericArgs<float>&, int, size_t, hipStream_t);
template int launch_frontend_generic_lists<double>(const FeModel*, const FeGenericArgs<double>&, int, size_t, hipStream_t);

}  // namespace irlosc
