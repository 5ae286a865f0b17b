// Translation unit of the wave-per-instance rigid-body front end (any tree of hinges), float and double records.
#include "osc_frontend.hpp"
#include "launchers.hpp"

namespace irlosc {

template <typename TOUT>
int launch_frontend_generic(const FeModel* dmodel, const double* qpos, const double* qvel, const FeOut<TOUT>& out, int B, size_t smem,
                            hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(osc_frontend_kernel<TOUT>, dim3(B < (1 << 20) ? B : (1 << 20)), dim3(64), smem, st, dmodel, qpos, qvel, out, B);
    return (int)hipGetLastError();
}
template int launch_frontend_generic<float>(const FeModel*, const double*, const double*, const FeOut<float>&, int, size_t, hipStream_t);
template int launch_frontend_generic<double>(const FeModel*, const double*, const double*, const FeOut<double>&, int, size_t, hipStream_t);

}  // namespace irlosc
