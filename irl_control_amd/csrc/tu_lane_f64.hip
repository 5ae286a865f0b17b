// Lane-per-robot OSC step of the fused path, double records (see tu_lane_impl.hpp); also the host-side plan of a layout
#define IRLOSC_LANE_TIN double
#define IRLOSC_LANE_PLAN
#include "tu_lane_impl.hpp"
