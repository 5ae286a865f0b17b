#pragma once
// Body of the translation units of the lane-per-robot OSC step of the fused path (osc_lane.hpp) on double records: the Dual-UR5 shapes with an
// instantiation (rows per end-effector body: stand, right arm, left arm) and the eigen pass behind them.
#include <algorithm>
#include <cstring>

#include "osc_lane.hpp"
#include "topo_dual_ur5.hpp"
#include "launchers.hpp"

namespace irlosc {

using lane::Shape;

template <class SH, typename TIN>
static int lane_launch(const Row16Train<TIN>& tr, const lane::LaneTrain& lt, int nsteps, int eig_blocks, int lane_min, hipStream_t st) {
    const KParams<TIN>& p = tr.p[0];
    const int waves = (p.B + 63) / 64;
    hipLaunchKernelGGL((lane::osc_lane_kernel<TopoDualUr5, SH, TIN>), dim3(waves, nsteps), dim3(64), 0, st, tr, lt);
    lane::EigTrain et;
    memset(&et, 0, sizeof et);
    et.B = p.B;
    for (int i = 0; i < nsteps; ++i)
        et.s[i] = lane::EigStep{tr.p[i].u, tr.p[i].flags, tr.x[i].worklist, tr.x[i].workcount, lt.rec[i], lt.rec_count[i]};
    et.lane_min = lane_min;
    // both forms of the eigen pass behind every lane kernel; each looks at the step's count of flagged robots and one of them returns
    // (grids no larger than the lists they can be handed: an idle launch is a few microseconds of a small batch's train)
    const int g_lane = std::min(eig_blocks, (p.B + 63) / 64);
    const int g_r16 = std::min(eig_blocks, (std::min(p.B, std::max(lane_min, 1)) + 3) / 4);
    if (p.B >= lane_min)
        hipLaunchKernelGGL((lane::osc_lane_eigen_kernel<TopoDualUr5, SH, TIN>), dim3(g_lane, nsteps), dim3(64), 0, st, et);
    if (lane_min > 0)
        hipLaunchKernelGGL((lane::osc_lane_eigen16_kernel<TopoDualUr5, SH, TIN>), dim3(g_r16, nsteps), dim3(64), 0, st, et);
    return (int)hipGetLastError();
}

#ifndef IRLOSC_LANE_TIN
#error "define IRLOSC_LANE_TIN (tu_lane_f64.hip / tu_lane_f32.hip)"
#endif

// tier = index into lane_tiers(): 0: (1, 6, 6)  1: (1, 3, 3)
template <>
int launch_lane_osc<IRLOSC_LANE_TIN>(const Row16Train<IRLOSC_LANE_TIN>& tr, const lane::LaneTrain& lt, int nsteps, int tier, int eig_blocks,
                                     int lane_min, hipStream_t st) {
    if (tr.p[0].B <= 0 || nsteps <= 0) return 0;
    switch (tier) {
        case 0: return lane_launch<Shape<1, 6, 6>, IRLOSC_LANE_TIN>(tr, lt, nsteps, eig_blocks, lane_min, st);
#ifndef IRLOSC_LANE_ONLY_TIER0      // (register / ISA experiments on one instantiation)
        case 1: return lane_launch<Shape<1, 3, 3>, IRLOSC_LANE_TIN>(tr, lt, nsteps, eig_blocks, lane_min, st);
#endif
        default: return (int)hipErrorNotSupported;
    }
}

#ifdef IRLOSC_LANE_PLAN
// does the lane kernel compute part 1 of the task signal itself (then no task pass is launched ahead of it)?
int lane_task_in_kernel() { return IRLOSC_LANE_TASK_IN; }

// Which instantiation takes a layout, and its row map: the task rows grouped by end-effector body (candidates of the compiled tree in
// body order), each group padded to the tier's rows.  -1: no instantiation (an EE body that is not a candidate cannot happen on a
// model that matched the tree; more rows on one body than any tier holds can: e.g. a base asked for three rotations) -- the fused
// path then keeps the row16 FROMQ kernel.
int lane_plan(const FeModel& h, lane::RowMap* map) {
    int cand[3], nc = 0;
    for (int b = 0; b < TopoDualUr5::NB; ++b) if (TopoDualUr5::ee_cand[b]) { if (nc < 3) cand[nc] = b; ++nc; }
    if (nc != 3) return -1;
    int cnt[3] = {0, 0, 0}, comp[3][IRLOSC_MAX_K], ext[3][IRLOSC_MAX_K], dev[3][IRLOSC_MAX_K];
    for (int d = 0; d < h.ndev; ++d) {
        int ci = -1;
        for (int i = 0; i < 3; ++i) if (cand[i] == h.ee_body[d]) ci = i;
        if (ci < 0) return -1;
        int row = h.row0[d];
        for (int i = 0; i < 6; ++i) {
            if (!((h.dofmask[d] >> i) & 1u)) continue;
            if (cnt[ci] >= IRLOSC_MAX_K) return -1;
            comp[ci][cnt[ci]] = i; ext[ci][cnt[ci]] = row; dev[ci][cnt[ci]] = d;
            ++cnt[ci]; ++row;
        }
    }
    int tier = -1;
    for (int t = lane::N_TIERS - 1; t >= 0 && tier < 0; --t)      // (tiers are listed from the largest to the smallest)
        if (cnt[0] <= lane::TIER_ROWS[t][0] && cnt[1] <= lane::TIER_ROWS[t][1] && cnt[2] <= lane::TIER_ROWS[t][2]) tier = t;
    if (tier < 0) return -1;
    memset(map, 0, sizeof *map);
    int r = 0;
    for (int ci = 0; ci < 3; ++ci)
        for (int s = 0; s < lane::TIER_ROWS[tier][ci]; ++s, ++r)
            if (s < cnt[ci]) {
                map->real |= 1u << r; map->comp[r] = comp[ci][s]; map->ext[r] = ext[ci][s]; map->dev[r] = dev[ci][s];
                map->canon[ext[ci][s]] = r;
            }
    for (int d = 0; d < h.ndev; ++d) map->ee_e0[d] = FeTopo<TopoDualUr5>::ee_index(h.ee_body[d]);
    return tier;
}
#endif

}  // namespace irlosc
