// Lane-per-robot OSC step of the fused path, float records (targets, gains, outputs; the exchange buffer is double either way)
#define IRLOSC_LANE_TIN float
#include "tu_lane_impl.hpp"
