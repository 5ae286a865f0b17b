// Translation unit of the lane-per-robot rigid-body front end with the Dual-UR5 tree shape compiled in, double records
// (one unit per record type: each instantiation is ~13 k fp64 instructions of straight-line code, ~100 s of compile time).
#include "osc_frontend_lane.hpp"
#include "topo_dual_ur5.hpp"
#include "launchers.hpp"

namespace irlosc {

template <typename TOUT>
int launch_frontend_lane_dual_ur5(const FeModel* dmodel, const double* qpos, const double* qvel, const FeOut<TOUT>& out, int B,
                                  double* side, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL((osc_frontend_lane_kernel<TopoDualUr5, TOUT>), dim3((B + 63) / 64), dim3(64), 0, st, dmodel, qpos, qvel, out, B, side);
    return (int)hipGetLastError();
}
template int launch_frontend_lane_dual_ur5<double>(const FeModel*, const double*, const double*, const FeOut<double>&, int, double*, hipStream_t);

int launch_frontend_lane_compact_dual_ur5(const FeModel* dmodel, const FeLaneTrain& tr, int nsteps, hipStream_t st) {
    if (tr.B <= 0 || nsteps <= 0) return 0;
    hipLaunchKernelGGL((osc_frontend_lane_compact_kernel<TopoDualUr5>), dim3((tr.B + 63) / 64, nsteps), dim3(64), 0, st, dmodel, tr);
    return (int)hipGetLastError();
}
int launch_q_layout(const double* qpos, const double* qvel, double* qt, int B, int nj, hipStream_t st) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(osc_q_layout_kernel<double>, dim3((B + 63) / 64), dim3(64), 0, st, qpos, qvel, qt, B, nj);
    return (int)hipGetLastError();
}
void frontend_lane_dual_ur5_tables(const FeModel& h, FeCompactTables* t) { frontend_lane_tables<TopoDualUr5>(h, t); }

size_t frontend_lane_dual_ur5_side_doubles_per_wave() { return (size_t)FeTopo<TopoDualUr5>::n_side() * 64; }
bool frontend_lane_dual_ur5_matches(const FeModel& h) { return frontend_lane_matches<TopoDualUr5>(h); }

}  // namespace irlosc
