// Internal constants shared by the .hip translation units; the public C ABI is include/bts_hip.h.
#pragma once
#include "../../include/bts_hip.h"

// groupnorm.hip: mean / rstd from per-block (sum, sumsq) partials laid out [N*G][B][2] (slab semantics), fixed order
int bts_gn_finalize_partials_(const double* partial, float* mean, float* rstd, int NG, long B, double count, float eps,
                              hipStream_t stream);
