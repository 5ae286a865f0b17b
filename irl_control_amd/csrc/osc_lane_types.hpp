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

// Records of the robots handed to the eigen pass: transposed in groups of 64 ([group][entry][64] doubles; lane::Rec in osc_lane.hpp has the
// entries of an instantiation -- 191 for rows (1, 6, 6): A's lower triangle 91, w 13, J's movable entries 85, robot index, row mask).
// What the host allocates per robot (whole groups):
constexpr int REC_DOUBLES = 192;

struct LaneTrain {
    const double* qt[R16_TRAIN];        // walk layout of the coordinates: [walk wave][2 NJ][64 robots] (entry 2 j + 1 = qvel_j)
    double* rec[R16_TRAIN];             // records of the step's flagged robots
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
    int32_t lane_min;                   // flagged robots of a step from which the lane-form pass takes it (below: the row16-form pass)
};

// tiers of canonical rows per end-effector body of the Dual-UR5 (stand dummy, right EE, left EE): launch_lane_osc's `tier`
constexpr int N_TIERS = 2;
constexpr int TIER_ROWS[N_TIERS][3] = {{1, 6, 6}, {1, 3, 3}};

}  // namespace lane
}  // namespace irlosc
