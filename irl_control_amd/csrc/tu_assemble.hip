// Translation unit of the state-assembly gather kernel, float and double.
#include "osc_assemble.hpp"
#include "launchers.hpp"

namespace irlosc {

template <typename T>
int launch_assemble(const RawDesc& d, const RawPtrs<T>& r, int B, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(osc_assemble_kernel<T>, dim3(B < 65536 ? B : 65536), dim3(64), 0, st, d, r, B);
    return (int)hipGetLastError();
}
template int launch_assemble<float>(const RawDesc&, const RawPtrs<float>&, int, hipStream_t);
template int launch_assemble<double>(const RawDesc&, const RawPtrs<double>&, int, hipStream_t);

template <typename T>
int launch_symmetry_probe(const T* M, int n, int B, int32_t* out, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(osc_symmetry_kernel<T>, dim3(B < 16384 ? B : 16384), dim3(64), 0, st, M, n, B, out);
    return (int)hipGetLastError();
}
template int launch_symmetry_probe<float>(const float*, int, int, int32_t*, hipStream_t);
template int launch_symmetry_probe<double>(const double*, int, int, int32_t*, hipStream_t);

template <typename T>
int launch_structure_probe(const T* M, const T* J, int n, int k, int B, const StructureMasks& m, int32_t* out, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(osc_structure_kernel<T>, dim3(B < 16384 ? B : 16384), dim3(64), 0, st, M, J, n, k, B, m, out);
    return (int)hipGetLastError();
}
template int launch_structure_probe<float>(const float*, const float*, int, int, int, const StructureMasks&, int32_t*, hipStream_t);
template int launch_structure_probe<double>(const double*, const double*, int, int, int, const StructureMasks&, int32_t*, hipStream_t);

}  // namespace irlosc
