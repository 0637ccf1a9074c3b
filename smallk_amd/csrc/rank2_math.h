// smallk_amd/csrc/rank2_math.h -- the closed-form rank-2 non-negative solve shared by the RANK2 kernels
// (rank2.hip: one launch per step; rank2_persist.hip: the whole factorisation in one launch).
// SystemSolveH nmf_solver_rank2.hpp:25-135 / SystemSolveW :139-212 (one fast Givens rotation, cosine or sine branch)
// followed by OptimalActiveSetH/W :216-318.  The formulas are the reference's, operation for operation.
#pragma once
#include "devutil.h"

namespace smk {

struct R2Solve {
    double t, b2, inv_a2, inv_d2, inv0, inv1, sq0, sq1;
    bool cosine, bad;
};

// what depends on the 2 x 2 left-hand side only (side 0: H from W'W, side 1: W from HH')
__device__ __forceinline__ R2Solve r2_prepare(double a00, double a01, double a11, int side)
{
    const double eps = DBL_EPSILON;
    const double a10 = a01;                                      // the Gram matrix is symmetric
    R2Solve s;
    s.bad = (fabs(a00) < eps) && (fabs(a01) < eps);              // "singular matrix"
    s.cosine = fabs(a00) >= fabs(a01);
    double a2, d2;
    if (side == 0) {
        if (s.cosine) { s.t = -a10 / a00; a2 = a00 - s.t * a10; s.b2 = a01 - s.t * a11; d2 = a11 + s.t * a01; }
        else          { s.t = -a00 / a10; a2 = -a10 + s.t * a00; s.b2 = -a11 + s.t * a01; d2 = a01 + s.t * a11; }
    } else {
        if (s.cosine) { s.t = a01 / a00; a2 = a00 + s.t * a01; s.b2 = a10 + s.t * a11; d2 = a11 - s.t * a10; }
        else          { s.t = a00 / a01; a2 = -a01 - s.t * a00; s.b2 = -a11 - s.t * a10; d2 = a10 - s.t * a11; }
    }
    s.inv_a2 = 1.0 / a2;
    s.inv_d2 = 1.0 / d2;
    if (fabs(d2 / a2) < eps) s.bad = true;
    s.inv0 = 1.0 / a00; s.inv1 = 1.0 / a11; s.sq0 = sqrt(a00); s.sq1 = sqrt(a11);
    return s;
}

// one right-hand side (b0, b1) -> the non-negative solution (x0, x1)
__device__ __forceinline__ void r2_apply(const R2Solve& s, int side, double b0, double b1, double& x0, double& x1)
{
    double e2, f2;
    if (side == 0) {
        if (s.cosine) { e2 = b0 - s.t * b1; f2 = b1 + s.t * b0; }
        else          { e2 = -b1 + s.t * b0; f2 = b0 + s.t * b1; }
    } else {
        if (s.cosine) { e2 = b0 + s.t * b1; f2 = b1 - s.t * b0; }
        else          { e2 = -b1 - s.t * b0; f2 = b0 - s.t * b1; }
    }
    x1 = f2 * s.inv_d2;
    x0 = (e2 - s.b2 * x1) * s.inv_a2;
    if (x0 <= 0.0 || x1 <= 0.0) {               // OptimalActiveSet
        double v1 = b0 * s.inv0, v2 = b1 * s.inv1;
        if (v1 * s.sq0 >= v2 * s.sq1) v2 = 0.0; else v1 = 0.0;
        x0 = v1;
        x1 = v2;
    }
}

}  // namespace smk
