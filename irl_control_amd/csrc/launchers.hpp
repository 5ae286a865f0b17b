// Host-side launch functions of the kernels, one translation unit per kernel family (tu_*.hip) so that the library builds
// in parallel and a change to one kernel recompiles one unit.  irlosc.hip (the C ABI) only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "osc_common.hpp"

namespace irlosc {

// tu_generic.hip -- one wavefront per instance (osc_generic.hpp); T = float, double
template <typename T>
int launch_generic(const KParams<T>& p, int blocks, hipStream_t st);

// tu_row16_f64.hip / tu_row16_f32.hip -- fp64-arithmetic row16 path (osc_row16.hpp); TIN = record type
template <typename TIN> struct Row16Train;
template <typename TIN>
int launch_row16(const Row16Train<TIN>& tr, int nsteps, bool tree, hipStream_t st);      // tree: tree-structured factorisation
void row16_tree_masks(uint32_t mrow[32], uint32_t* jcols);      // zero pattern the tree form relies on (osc_row16.hpp)
// parts: bit 0 = the task pass (part 1 of the task signal as rows of the exchange buffer), bit 1 = the row16 FROMQ kernel
template <typename TIN>
int launch_row16_fromq(const Row16Train<TIN>& tr, int nsteps, hipStream_t st, int parts = 3);
template <typename TIN>
int launch_row16_worklist(const Row16Train<TIN>& tr, int nsteps, int32_t* reset, hipStream_t st);
// tu_row16_pad_{dense,tree,fromq}_{f64,f32}.hip -- the KMAX-padded variants launch_row16 / launch_row16_fromq fall through to for the
// layouts without an instantiation of their own (explicit specialisations, one translation unit each)
template <typename TIN> int launch_row16_pad_dense(const Row16Train<TIN>& tr, int nsteps, hipStream_t st);
template <typename TIN> int launch_row16_pad_tree(const Row16Train<TIN>& tr, int nsteps, hipStream_t st);
template <typename TIN> int launch_row16_pad_fromq(const Row16Train<TIN>& tr, int nsteps, hipStream_t st, int parts);
template <> int launch_row16_pad_dense<double>(const Row16Train<double>&, int, hipStream_t);
template <> int launch_row16_pad_dense<float>(const Row16Train<float>&, int, hipStream_t);
template <> int launch_row16_pad_tree<double>(const Row16Train<double>&, int, hipStream_t);
template <> int launch_row16_pad_tree<float>(const Row16Train<float>&, int, hipStream_t);
template <> int launch_row16_pad_fromq<double>(const Row16Train<double>&, int, hipStream_t, int);
template <> int launch_row16_pad_fromq<float>(const Row16Train<float>&, int, hipStream_t, int);

// tu_lane_f64.hip / tu_lane_f32.hip -- the OSC step of the fused path in lane-per-robot form + its eigen pass (osc_lane.hpp).
// tier: 0 = rows per end-effector body (stand, right, left) up to (1, 6, 6), 1 = up to (1, 3, 3)
namespace lane { struct LaneTrain; struct RowMap; }
struct FeModel;
int lane_plan(const FeModel& h, lane::RowMap* map);      // -> tier, or -1: no instantiation for this layout
int lane_task_in_kernel();                              // 1: the lane kernel computes part 1 of the task signal itself (no task pass)
template <typename TIN>
int launch_lane_osc(const Row16Train<TIN>& tr, const lane::LaneTrain& lt, int nsteps, int tier, int eig_blocks, int lane_min, hipStream_t st);
template <> int launch_lane_osc<double>(const Row16Train<double>&, const lane::LaneTrain&, int, int, int, int, hipStream_t);
template <> int launch_lane_osc<float>(const Row16Train<float>&, const lane::LaneTrain&, int, int, int, int, hipStream_t);

// tu_frontend.hip / tu_frontend_lane.hip -- rigid-body front end (osc_frontend.hpp, osc_frontend_lane.hpp); TOUT = record type
struct FeModel;
template <typename TOUT> struct FeOut;
template <typename TOUT>
int launch_frontend_generic(const FeModel* dmodel, const double* qpos, const double* qvel, const FeOut<TOUT>& out, int B, size_t smem,
                            hipStream_t st);
template <typename TOUT>
int launch_frontend_lane_dual_ur5(const FeModel* dmodel, const double* qpos, const double* qvel, const FeOut<TOUT>& out, int B,
                                  double* side, hipStream_t st);
template <typename TOUT> struct FeGenericArgs;
template <typename TOUT>
int launch_frontend_generic_lists(const FeModel* dmodel, const FeGenericArgs<TOUT>& a, int nsteps, size_t smem, hipStream_t st);
// compact form for the fused path: a train of up to FE_TRAIN steps, nothing but the exchange buffer is written
struct FeLaneTrain;
struct FeCompactTables;
int launch_frontend_lane_compact_dual_ur5(const FeModel* dmodel, const FeLaneTrain& tr, int nsteps, hipStream_t st);
// tu_frontend_lane_compact_s.hip -- the same walk with the structural constants of the Dual-UR5's MJCF compiled in (TopoDualUr5S)
int launch_frontend_lane_compact_dual_ur5_s(const FeModel* dmodel, const FeLaneTrain& tr, int nsteps, hipStream_t st);
bool frontend_lane_dual_ur5_s_matches(const FeModel& h);
int launch_q_layout(const double* qpos, const double* qvel, double* qt, int B, int nj, hipStream_t st);      // [wave][2 nj][64]: the fused walk's input layout
void frontend_lane_dual_ur5_tables(const FeModel& h, FeCompactTables* t);
size_t frontend_lane_dual_ur5_side_doubles_per_wave();
bool frontend_lane_dual_ur5_matches(const FeModel& h);

// tu_assemble.hip -- state assembly from raw simulator arrays (osc_assemble.hpp)
struct RawDesc;
template <typename T> struct RawPtrs;
template <typename T>
int launch_assemble(const RawDesc& d, const RawPtrs<T>& r, int B, hipStream_t st);
// out[0] += instances whose M is not symmetric, out[1] = min(out[1], first such instance)
template <typename T>
int launch_symmetry_probe(const T* M, int n, int B, int32_t* out, hipStream_t st);
// out[0] += instances with a non-zero where the masks say zero (M[j][c] outside mrow[j], a column of J outside jcols)
template <typename T>
int launch_structure_probe(const T* M, const T* J, int n, int k, int B, const StructureMasks& m, int32_t* out, hipStream_t st);

}  // namespace irlosc
