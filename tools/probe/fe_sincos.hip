// Probe: accuracy of the inline sin / cos of the lane-per-robot front end (csrc/osc_frontend_lane.hpp: fe_sincos) against
// the host libm, in ulps, by magnitude of the argument.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on tools/probe/fe_sincos.hip -o tools/probe/fe_sincos && tools/probe/fe_sincos
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <type_traits>
#include <utility>
#include <vector>
#include "../../irl_control_amd/csrc/osc_frontend_lane.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(const double* x, double* s, double* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) irlosc::fe_sincos(x[i], s[i], c[i]);
}

static double ulps(double got, double want) {
    if (got == want) return 0.0;
    const double u = std::nextafter(std::fabs(want), INFINITY) - std::fabs(want);
    return std::fabs(got - want) / u;
}

int main() {
    const int n = 1 << 20;
    std::mt19937_64 rng(7);
    const double hi[] = {0.8, 10.0, 1e3, 1e5, 1e6, 1e8};
    double *dx, *ds, *dc;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&ds, n * 8)); CK(hipMalloc(&dc, n * 8));
    std::vector<double> x(n), s(n), c(n);
    double lo = 0.0;
    for (double h : hi) {
        std::uniform_real_distribution<double> d(lo, h);
        for (int i = 0; i < n; ++i) x[i] = (i & 1 ? -1.0 : 1.0) * d(rng);
        CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
        CK(hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost));
        double ms = 0, mc = 0, as = 0, ac = 0;
        for (int i = 0; i < n; ++i) {
            const long double xl = x[i];
            const double ws = (double)sinl(xl), wc = (double)cosl(xl);
            ms = std::fmax(ms, ulps(s[i], ws)); mc = std::fmax(mc, ulps(c[i], wc));
            as = std::fmax(as, std::fabs(s[i] - ws)); ac = std::fmax(ac, std::fabs(c[i] - wc));
        }
        printf("|x| in [%g, %g): sin max %.2f ulp (abs %.2e), cos max %.2f ulp (abs %.2e)\n", lo, h, ms, as, mc, ac);
        lo = h;
    }
    return 0;
}
