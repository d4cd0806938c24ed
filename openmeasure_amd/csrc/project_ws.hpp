// W-stationary projection kernel (project_ws.hip), internal to the library: tried first by project.hip's entry points.
#pragma once
#include "common.hpp"

// SPR_OK when launched; SPR_E_UNSUPPORTED when the shape is outside its range (m not 64/128/192/256 packed and 16-byte
// aligned, r > 64, accumulate) -- the caller then launches the general kernel.  Other codes are errors.
// d_rownorm2 != NULL: the squared norms of the stored rows of Ur are written there as well (n_rows doubles).
template <typename TX, typename TU>
int spr_project_ws(const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
                   int32_t n_features, int32_t center, const double *d_inv_scale, const double *d_rowmean,
                   const double *d_W, int32_t r, TU *d_Ur, int64_t ldu, int32_t accumulate,
                   double *d_rownorm2, hipStream_t st);
