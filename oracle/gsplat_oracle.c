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

/* F3  Gaussians::Initialize   src/gaussian.cpp:38-104   ("next" row f3), type independent: coordinates are doubles.
 * Brute-force exact kNN: the k+1 smallest squared distances including the point itself, first one dropped, mean of
 * the square roots of the rest; 0.01 when the point has no neighbour. */
void orc_knn_mean_distance(const double *pts, long n, int k, float *out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (long i = 0; i < n; ++i) {
    double best[9];
    const int want = k + 1;
    for (int a = 0; a < want; ++a) best[a] = INFINITY;
    for (long q = 0; q < n; ++q) {
      const double dx = pts[3 * q] - pts[3 * i], dy = pts[3 * q + 1] - pts[3 * i + 1], dz = pts[3 * q + 2] - pts[3 * i + 2];
      const double d = dx * dx + dy * dy + dz * dz;
      if (!(d < best[want - 1])) continue;
      int a = want - 1;
      while (a > 0 && best[a - 1] > d) { best[a] = best[a - 1]; --a; }
      best[a] = d;
    }
    double total = 0.0;
    int count = 0;
    for (int a = 1; a < want; ++a)
      if (best[a] < INFINITY) { total += sqrt(best[a]); ++count; }
    out[i] = count > 0 ? (float)(total / count) : 0.01f;
  }
}

/* attribute part of Gaussians::Initialize (src/gaussian.cpp:93-101) */
void orc_init_attributes(const double *pts, const unsigned char *colors, const float *mean_dist, long n, float *xyz,
                         float *rgb, float *opacity, float *scale, float *quat) {
  const float C0 = 0.28209479177387814f;
  for (long i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) {
      xyz[3 * i + a] = (float)pts[3 * i + a];
      rgb[3 * i + a] = ((float)colors[3 * i + a] / 255.0f - 0.5f) / C0;
      scale[3 * i + a] = logf(mean_dist[i]);
    }
    opacity[i] = logf(0.2f) - logf(1.0f - 0.2f);
    quat[4 * i] = 1.0f; quat[4 * i + 1] = quat[4 * i + 2] = quat[4 * i + 3] = 0.0f;
  }
}

int orc_version(void) { return 1; }
