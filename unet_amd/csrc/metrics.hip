// Loss / metric reductions (placeholder translation unit; kernels are added with their ABI).
#include "common.h"
