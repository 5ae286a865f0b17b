// KMAX-padded row16 kernels, float records, fromq form (see tu_row16_pad_impl.hpp)
#define IRLOSC_PAD_TIN float
#define IRLOSC_PAD_FROMQ
#include "tu_row16_pad_impl.hpp"
