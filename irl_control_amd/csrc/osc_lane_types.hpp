// Plain types shared by the lane-per-robot OSC kernels (osc_lane.hpp) and the C ABI layer (irlosc.hip).
#pragma once
#include <stdint.h>

#include "osc_row16.hpp"      // R16_TRAIN

namespace irlosc {
namespace lane {

// What the host derives from the layout (irlosc_set_model): uniform over the launch, read as scalars
struct RowMap {
    uint32_t real;                  // bit r: canonical row r is a task row of the layout (else padding)
    int32_t comp[IRLOSC_MAX_K];     // component 0..5 of (jacp, jacr) the row takes (device.py:131-132); 0 for padding
    int32_t ext[IRLOSC_MAX_K];      // its row in targets order (the task pass's row, osc.py:134-138); 0 for padding
    int32_t dev[IRLOSC_MAX_K];      // its target device; 0 for padding
    int32_t canon[IRLOSC_MAX_K];    // the other way round: task row (targets order) -> canonical row
    int32_t ee_e0[IRLOSC_MAX_DEV];  // target device -> first entry of its end effector's pose in the exchange block (x y z qw qx qy qz)
};

// Records of the robots handed to the eigen pass, in doubles
constexpr int REC_A = 0;            // [r][16]: A[r][c]
constexpr int REC_W = 256;          // [16]
constexpr int REC_J = 272;          // [r][16]: J[r][EE hinge of rank c]
constexpr int REC_META = 528;       // +0: robot index, +1: bit r = row r is padding or an exact zero row (both as integers in the double's bits)
constexpr int REC_DOUBLES = 544;

struct LaneTrain {
    const double* qt[R16_TRAIN];        // walk layout of the coordinates: [walk wave][2 NJ][64 robots] (entry 2 j + 1 = qvel_j)
    double* rec[R16_TRAIN];             // records of the step's flagged robots, REC_DOUBLES each
    int32_t* rec_count[R16_TRAIN];      // zero on entry
    RowMap map;
};

// What the eigen pass needs of a step, and nothing else (the two trains of the lane kernel are 3 kB of kernel arguments; indexed by
// blockIdx.y they cost the pass a hundred spilled scalar registers)
struct EigStep {
    void* u;                            // [B][n] torques, record type
    uint32_t* flags;                    // [B]
    int32_t* worklist;                  // give-up list of the step (-> generic kernel)
    int32_t* workcount;
    const double* rec;
    const int32_t* rec_count;
};
struct EigTrain {
    EigStep s[R16_TRAIN];
    int32_t B;
};

// tiers of canonical rows per end-effector body of the Dual-UR5 (stand dummy, right EE, left EE): launch_lane_osc's `tier`
constexpr int N_TIERS = 2;
constexpr int TIER_ROWS[N_TIERS][3] = {{1, 6, 6}, {1, 3, 3}};

}  // namespace lane
}  // namespace irlosc
