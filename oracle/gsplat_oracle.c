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

/* OpenMP threads of the per-gaussian operators (P1 P2 K1 G1 G2 S1 and the backward chain): every gaussian is
 * independent, so the arithmetic is the same at any count.  The compositing functions take their own `threads`
 * argument; tile binning (the candidate scan and the qsort) is serial at every setting.  Default 1. */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 0 ? n : 1; }

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

/* The same quantity the way the reference computes it (src/gaussian.cpp:60-91): a kd-tree over the points (nanoflann
 * KDTreeSingleIndexAdaptor, leaf size 10, L2 in double) and one k+1-nearest query per point under OpenMP.  nanoflann is
 * not vendored in the reference (empty submodule), so this is a dependency-free restatement of the published
 * algorithm -- median split on the widest axis, leaves of <= 10 points, depth-first search that visits the near child
 * first and the far child only if the splitting plane is closer than the current k-th distance -- used as the CPU
 * baseline of the initialisation (bench.py) and checked against the brute force above. */
typedef struct { int lo, hi, axis, left, right; double split; } orc_kdnode;
typedef struct { const double *pts; int *idx; orc_kdnode *nodes; int n_nodes; } orc_kdtree;

static int orc_kd_build(orc_kdtree *t, int lo, int hi) {
  const int me = t->n_nodes++;
  orc_kdnode *nd = &t->nodes[me];
  nd->lo = lo; nd->hi = hi; nd->left = nd->right = -1; nd->axis = 0; nd->split = 0.0;
  if (hi - lo <= 10) return me;
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = lo; k < hi; ++k)
    for (int a = 0; a < 3; ++a) {
      const double v = t->pts[3 * (long)t->idx[k] + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int ax = 0;
  for (int a = 1; a < 3; ++a)
    if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
  if (!(mx[ax] > mn[ax])) return me;  /* all points identical: one (large) leaf */
  /* nth_element on the axis (quickselect, median) */
  int l = lo, r = hi - 1;
  const int mid = lo + (hi - lo) / 2;
  while (l < r) {
    const double pivot = t->pts[3 * (long)t->idx[(l + r) / 2] + ax];
    int i = l, j = r;
    while (i <= j) {
      while (t->pts[3 * (long)t->idx[i] + ax] < pivot) ++i;
      while (t->pts[3 * (long)t->idx[j] + ax] > pivot) --j;
      if (i <= j) { const int tmp = t->idx[i]; t->idx[i] = t->idx[j]; t->idx[j] = tmp; ++i; --j; }
    }
    if (mid <= j) r = j; else if (mid >= i) l = i; else break;
  }
  nd->axis = ax;
  nd->split = t->pts[3 * (long)t->idx[mid] + ax];
  const int left = orc_kd_build(t, lo, mid);
  const int right = orc_kd_build(t, mid, hi);
  t->nodes[me].left = left; t->nodes[me].right = right;  /* nodes[] does not move: sized up front */
  return me;
}

static void orc_kd_query(const orc_kdtree *t, int node, const double *q, double *best, int want) {
  const orc_kdnode *nd = &t->nodes[node];
  if (nd->left < 0) {
    for (int k = nd->lo; k < nd->hi; ++k) {
      const double *p = &t->pts[3 * (long)t->idx[k]];
      const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2], d = dx * dx + dy * dy + dz * dz;
      if (!(d < best[want - 1])) continue;
      int a = want - 1;
      while (a > 0 && best[a - 1] > d) { best[a] = best[a - 1]; --a; }
      best[a] = d;
    }
    return;
  }
  const double diff = q[nd->axis] - nd->split;
  const int near = diff < 0.0 ? nd->left : nd->right, far = diff < 0.0 ? nd->right : nd->left;
  orc_kd_query(t, near, q, best, want);
  if (diff * diff < best[want - 1]) orc_kd_query(t, far, q, best, want);
}

void orc_knn_mean_distance_kdtree(const double *pts, long n, int k, float *out, int threads) {
  if (n <= 0) return;
  orc_kdtree t;
  t.pts = pts;
  t.idx = (int *)malloc((size_t)n * sizeof(int));
  t.nodes = (orc_kdnode *)malloc((size_t)(2 * n + 1) * sizeof(orc_kdnode));
  t.n_nodes = 0;
  for (long i = 0; i < n; ++i) t.idx[i] = (int)i;
  orc_kd_build(&t, 0, (int)n);
  const int want = k + 1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 256)
  for (long i = 0; i < n; ++i) {
    double best[9];
    for (int a = 0; a < want; ++a) best[a] = INFINITY;
    orc_kd_query(&t, 0, &pts[3 * i], best, want);
    double total = 0.0;
    int count = 0;
    for (int a = 1; a < want; ++a)
      if (best[a] < INFINITY) { total += sqrt(best[a]); ++count; }
    out[i] = count > 0 ? (float)(total / count) : 0.01f;
  }
  free(t.idx);
  free(t.nodes);
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

/* F4 operators.  compute_morton_codes: cuda/culling.cu:14-63 (the bit-spread masks are the reference's own). */
static unsigned long long orc_spread_bits(unsigned long long n) {
  n &= 0x1FFFFFull;
  n = (n | (n << 32)) & 0x1F000000FFFFull;
  n = (n | (n << 16)) & 0x1F0000FF0000FFull;
  n = (n | (n << 8)) & 0x100F807C0F807C0Full;
  n = (n | (n << 4)) & 0x1084210842108421ull;
  n = (n | (n << 2)) & 0x1249249249249249ull;
  return n;
}
static unsigned long long orc_to_u64(float v) {
  if (!(v > 0.0f)) return 0ull;
  if (v >= 18446744073709551616.0f) return ~0ull;
  return (unsigned long long)v;
}
void orc_morton_codes(long n, const float *xyz, float x_max, float y_max, float z_max, float x_min, float y_min,
                      float z_min, unsigned long long *codes) {
  const float kMax = 2097151.0f;
  for (long i = 0; i < n; ++i) {
    const unsigned long long xq = orc_to_u64((xyz[3 * i] - x_min) * (kMax / (x_max - x_min)));
    const unsigned long long yq = orc_to_u64((xyz[3 * i + 1] - y_min) * (kMax / (y_max - y_min)));
    const unsigned long long zq = orc_to_u64((xyz[3 * i + 2] - z_min) * (kMax / (z_max - z_min)));
    codes[i] = (orc_spread_bits(zq) << 2) | (orc_spread_bits(yq) << 1) | orc_spread_bits(xq);
  }
}

/* split_gaussians: cuda/adaptive_density.cu:68-164 with the build's counter-based normal generator
 * (3dgs_amd/csrc/gs_density.hip) in place of time-seeded cuRAND.  clone = the same copies without the draw. */
static unsigned long long orc_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static float orc_normal(unsigned long long seed, unsigned long long counter) {
  const unsigned long long bits = orc_splitmix64(orc_splitmix64(seed) ^ (counter * 0xD1342543DE82EF95ull + 1ull));
  const float u1 = (float)((bits >> 40) + 1ull) * (1.0f / 16777216.0f);
  const float u2 = (float)((bits >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
void orc_clone_split(long n, int split, float scale_factor, int num_sh_coef, const unsigned char *mask,
                     const int *write_ids, const float *xyz, const float *rgb, const float *op, const float *scale,
                     const float *quat, const float *sh, float *xyz_o, float *rgb_o, float *op_o, float *scale_o,
                     float *quat_o, float *sh_o, unsigned long long seed) {
  const long w = (long)num_sh_coef * 3;
  for (long i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    const int copies = split ? 2 : 1;
    const long base = (long)write_ids[i] * copies;
    float Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, e[3] = {0, 0, 0};
    if (split) {
      for (int a = 0; a < 3; ++a) e[a] = expf(scale[3 * i + a]);
      const float q0 = quat[4 * i], q1 = quat[4 * i + 1], q2 = quat[4 * i + 2], q3 = quat[4 * i + 3];
      const float inv = 1.0f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
      const float qw = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
      Rm[0] = 1.0f - 2.0f * (y * y + z * z); Rm[1] = 2.0f * (x * y - qw * z); Rm[2] = 2.0f * (x * z + qw * y);
      Rm[3] = 2.0f * (x * y + qw * z); Rm[4] = 1.0f - 2.0f * (x * x + z * z); Rm[5] = 2.0f * (y * z - qw * x);
      Rm[6] = 2.0f * (x * z - qw * y); Rm[7] = 2.0f * (y * z + qw * x); Rm[8] = 1.0f - 2.0f * (x * x + y * y);
    }
    for (int j = 0; j < copies; ++j) {
      const long d = base + j;
      float v[3] = {0, 0, 0};
      if (split) {
        const unsigned long long c = ((unsigned long long)i * 2ull + (unsigned long long)j) * 3ull;
        for (int a = 0; a < 3; ++a) v[a] = orc_normal(seed, c + a) * e[a];
      }
      for (int a = 0; a < 3; ++a) {
        xyz_o[3 * d + a] = xyz[3 * i + a] + (split ? (v[0] * Rm[3 * a] + v[1] * Rm[3 * a + 1] + v[2] * Rm[3 * a + 2]) : 0.0f);
        scale_o[3 * d + a] = split ? logf(e[a] / scale_factor) : scale[3 * i + a];
        rgb_o[3 * d + a] = rgb[3 * i + a];
      }
      op_o[d] = op[i];
      for (int a = 0; a < 4; ++a) quat_o[4 * d + a] = quat[4 * i + a];
      for (long k = 0; k < w; ++k) sh_o[d * w + k] = sh[i * w + k];
    }
  }
}

int orc_version(void) { return 1; }
