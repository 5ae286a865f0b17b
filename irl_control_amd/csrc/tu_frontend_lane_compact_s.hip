// Translation unit of the COMPACT walk of the fused path instantiated with the Dual-UR5 shape AND the structural constants of its
// MJCF (TopoDualUr5S, topo_dual_ur5.hpp): unrotated / coincident frames, hinges about coordinate axes anchored at their body's origin,
// diagonal body-frame inertias -- the products with those exact zeros and ones are compiled out (a fifth of the walk's arithmetic).
// irlosc_set_model selects it for a model that carries exactly these constants; any other Dual-UR5-shaped model runs the
// shape-only instantiation (tu_frontend_lane_f64.hip).
#include "osc_frontend_lane.hpp"
#include "topo_dual_ur5.hpp"
#include "launchers.hpp"

namespace irlosc {

int launch_frontend_lane_compact_dual_ur5_s(const FeModel* dmodel, const FeLaneTrain& tr, int nsteps, hipStream_t st) {
    if (tr.B <= 0 || nsteps <= 0) return 0;
    hipLaunchKernelGGL((osc_frontend_lane_compact_kernel<TopoDualUr5S>), dim3((tr.B + 63) / 64, nsteps), dim3(64), 0, st, dmodel, tr);
    return (int)hipGetLastError();
}
bool frontend_lane_dual_ur5_s_matches(const FeModel& h) { return frontend_lane_matches<TopoDualUr5S>(h); }

}  // namespace irlosc
