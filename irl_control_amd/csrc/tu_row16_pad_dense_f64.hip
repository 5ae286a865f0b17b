// KMAX-padded row16 kernels, double records, dense form (see tu_row16_pad_impl.hpp)
#define IRLOSC_PAD_TIN double
#define IRLOSC_PAD_DENSE
#include "tu_row16_pad_impl.hpp"
