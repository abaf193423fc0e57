// sc_sweep_tb.hip -- temporally blocked sweep kernels (placeholder until the fused kernels land).
#include "sc_common.h"
namespace sc {
bool launch_jacobi_tb(Field, Field, Field, int, hipStream_t) { return false; }
bool launch_rb_tb(Field, Field, Field, int, float, hipStream_t) { return false; }
} // namespace sc
