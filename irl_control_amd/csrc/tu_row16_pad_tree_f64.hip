// KMAX-padded row16 kernels, double records, tree form (see tu_row16_pad_impl.hpp)
#define IRLOSC_PAD_TIN double
#define IRLOSC_PAD_TREE
#include "tu_row16_pad_impl.hpp"
