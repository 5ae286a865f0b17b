// KMAX-padded row16 kernels, float records, dense form (see tu_row16_pad_impl.hpp)
#define IRLOSC_PAD_TIN float
#define IRLOSC_PAD_DENSE
#include "tu_row16_pad_impl.hpp"
