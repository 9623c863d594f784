/*
 * gsplat_oracle.c -- TEST INFRASTRUCTURE ONLY: the CPU parity oracle for the MI355X
 * gaussian-splat rasterizer.  See gsplat_oracle_impl.h for the per-function citations
 * into the reference (AndrewBoessen/3DGS, cuda/ tree) and oracle/README.md for how the
 * oracle is pinned.  Built by oracle/Makefile into oracle/liboracle.so.
 *
 * Two instantiations of the same source:
 *   *_f32  float32, -ffp-contract=off: the parity checker for the HIP kernels
 *   *_f64  float64: used only by tests to verify the backward formulas by finite
 *          differences of the forward ones (the reference's tests do the same in f32).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define R float
#define SUF _f32
#define SQRT sqrtf
#define EXP expf
#define CEIL ceilf
#define FLOOR floorf
#define ATAN2 atan2f
#define SIN sinf
#define COS cosf
#define FABS fabsf
#define FMIN fminf
#define FMAX fmaxf
#include "gsplat_oracle_impl.h"
#undef R
#undef SUF
#undef SQRT
#undef EXP
#undef CEIL
#undef FLOOR
#undef ATAN2
#undef SIN
#undef COS
#undef FABS
#undef FMIN
#undef FMAX

#define R double
#define SUF _f64
#define SQRT sqrt
#define EXP exp
#define CEIL ceil
#define FLOOR floor
#define ATAN2 atan2
#define SIN sin
#define COS cos
#define FABS fabs
#define FMIN fmin
#define FMAX fmax
#include "gsplat_oracle_impl.h"

int orc_version(void) { return 1; }
