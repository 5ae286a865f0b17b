// Translation unit of the generic kernel (one wavefront per instance, any n <= 32, k <= 16), float and double.
#include "osc_generic.hpp"
#include "launchers.hpp"

namespace irlosc {

template <typename T>
int launch_generic(const KParams<T>& p, int blocks, hipStream_t st) {
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(osc_generic_kernel<T>, dim3(blocks), dim3(64), generic_smem_bytes<T>(p.n, p.k, p.ndev), st, p);
    return (int)hipGetLastError();
}
template int launch_generic<float>(const KParams<float>&, int, hipStream_t);
template int launch_generic<double>(const KParams<double>&, int, hipStream_t);

}  // namespace irlosc
