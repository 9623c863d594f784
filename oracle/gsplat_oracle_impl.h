/*
 * gsplat_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * CPU restatement of the reference's differentiable tile rasterizer
 * (AndrewBoessen/3DGS, cuda/ tree).  It is included twice by gsplat_oracle.c, once with
 * R=float (the parity instantiation: same operation order as the reference kernels,
 * compiled with -ffp-contract=off) and once with R=double (used only to verify the
 * float backward formulas by finite differences).
 *
 * Nothing in the product path (3dgs_amd/, include/) may call into this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Parity pinning: see oracle/README.md -- the l<=2 SH basis and all
 * other operators are pinned by the reference's own known-answer tests
 * (tests/cuda_forward_test.cpp, tests/cuda_backward_test.cpp); the l=3 SH basis and the
 * per-coefficient SH gradients come from sphericart, which is NOT vendored in the
 * reference: they are pinned against scipy.special.sph_harm_y in sphericart's published
 * convention (tests/test_oracle_known_answers.py); only the sphericart VERSION the
 * reference would link stays unpinned.
 */

#ifndef R
#error "define R, SUF and the math macros before including"
#endif

#define FN2(a, b) a##b
#define FN1(a, b) FN2(a, b)
#define FN(name) FN1(name, SUF)

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

/* float -> int conversion with the device semantics (NaN -> 0, saturating). */
static inline int FN(f2i_sat)(R v) {
  if (v != v) return 0;
  if (v >= (R)2147483647.0) return 2147483647;
  if (v <= (R)-2147483648.0) return (-2147483647 - 1);
  return (int)v;
}

/* ------------------------------------------------------------------------------------ */
/* P1  compute_camera_space_points   cuda/projection.cu:6-45                             */
/* ------------------------------------------------------------------------------------ */
void FN(orc_camera_space_points)(const R *xyz_w, const R *view, int N, R *xyz_c) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R wx = xyz_w[i * 3 + 0], wy = xyz_w[i * 3 + 1], wz = xyz_w[i * 3 + 2];
    xyz_c[i * 3 + 0] = view[0] * wx + view[1] * wy + view[2] * wz + view[3];
    xyz_c[i * 3 + 1] = view[4] * wx + view[5] * wy + view[6] * wz + view[7];
    xyz_c[i * 3 + 2] = view[8] * wx + view[9] * wy + view[10] * wz + view[11];
  }
}

/* ------------------------------------------------------------------------------------ */
/* P2  project_to_screen   cuda/projection.cu:47-98  (row 2 of proj is never read)       */
/* ------------------------------------------------------------------------------------ */
void FN(orc_project_to_screen)(const R *xyz, const R *proj, int N, int width, int height, R *uv) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    const R x_clip = proj[0] * x + proj[1] * y + proj[2] * z + proj[3];
    const R y_clip = proj[4] * x + proj[5] * y + proj[6] * z + proj[7];
    const R w_clip = proj[12] * x + proj[13] * y + proj[14] * z + proj[15];
    const R x_ndc = x_clip / (w_clip + (R)1e-6f);
    const R y_ndc = y_clip / (w_clip + (R)1e-6f);
    uv[i * 2 + 0] = (x_ndc * (R)0.5 + (R)0.5) * (R)width;
    uv[i * 2 + 1] = (y_ndc * (R)0.5 + (R)0.5) * (R)height;
  }
}

/* ------------------------------------------------------------------------------------ */
/* K1  cull_gaussians   cuda/culling.cu:70-95   (true = keep, inclusive bounds)          */
/* ------------------------------------------------------------------------------------ */
void FN(orc_cull_gaussians)(const R *uv, const R *xyz, int N, R near_thresh, int padding, int width, int height,
                            unsigned char *mask) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R u = uv[i * 2 + 0], v = uv[i * 2 + 1], z = xyz[i * 3 + 2];
    const int zok = z >= near_thresh;
    const int fok = u >= (R)(-1 * padding) && u <= (R)(width + padding) && v >= (R)(-1 * padding) &&
                    v <= (R)(height + padding);
    mask[i] = (unsigned char)(zok && fok);
  }
}

/* ------------------------------------------------------------------------------------ */
/* G1  compute_sigma   cuda/gaussian.cu:6-75   quaternion order (w,x,y,z)               */
/* ------------------------------------------------------------------------------------ */
void FN(orc_compute_sigma)(const R *quaternion, const R *scale, int N, R *sigma) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    R w = quaternion[4 * i + 0], x = quaternion[4 * i + 1], y = quaternion[4 * i + 2], z = quaternion[4 * i + 3];
    const R norm = SQRT(w * w + x * x + y * y + z * z);
    const R inv_norm = (R)1 / (norm + (R)1e-6f);
    w *= inv_norm; x *= inv_norm; y *= inv_norm; z *= inv_norm;
    const R x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const R r00 = (R)1 - (R)2 * (y2 + z2), r01 = (R)2 * (xy - wz), r02 = (R)2 * (xz + wy);
    const R r10 = (R)2 * (xy + wz), r11 = (R)1 - (R)2 * (x2 + z2), r12 = (R)2 * (yz - wx);
    const R r20 = (R)2 * (xz - wy), r21 = (R)2 * (yz + wx), r22 = (R)1 - (R)2 * (x2 + y2);
    const R sx = EXP(scale[3 * i + 0]), sy = EXP(scale[3 * i + 1]), sz = EXP(scale[3 * i + 2]);
    const R rs00 = r00 * sx, rs10 = r10 * sx, rs20 = r20 * sx;
    const R rs01 = r01 * sy, rs11 = r11 * sy, rs21 = r21 * sy;
    const R rs02 = r02 * sz, rs12 = r12 * sz, rs22 = r22 * sz;
    R *s = sigma + 6 * i;
    s[0] = rs00 * rs00 + rs01 * rs01 + rs02 * rs02;
    s[1] = rs00 * rs10 + rs01 * rs11 + rs02 * rs12;
    s[2] = rs00 * rs20 + rs01 * rs21 + rs02 * rs22;
    s[3] = rs10 * rs10 + rs11 * rs11 + rs12 * rs12;
    s[4] = rs10 * rs20 + rs11 * rs21 + rs12 * rs22;
    s[5] = rs20 * rs20 + rs21 * rs21 + rs22 * rs22;
  }
}

/* ------------------------------------------------------------------------------------ */
/* G2  compute_conic = jacobian kernel (cuda/gaussian.cu:177-218) + conic kernel (:77-175) */
/* radius is float4 {r_major, r_minor, sin_theta, cos_theta} stored as 4 consecutive R.    */
/* ------------------------------------------------------------------------------------ */
void FN(orc_projection_jacobian)(const R *xyz, R focal_x, R focal_y, R tan_fovx, R tan_fovy, int N, R *J) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    R x = xyz[i * 3 + 0], y = xyz[i * 3 + 1];
    const R z = xyz[i * 3 + 2];
    R *j = J + 6 * i;
    if (FABS(z) < (R)1e-6f) {
      j[0] = j[1] = j[2] = j[3] = j[4] = j[5] = (R)0;
      continue;
    }
    const R limx = (R)1.3f * tan_fovx, limy = (R)1.3f * tan_fovy;
    const R txtz = x / z, tytz = y / z;
    x = FMIN(limx, FMAX(-limx, txtz)) * z;
    y = FMIN(limy, FMAX(-limy, tytz)) * z;
    j[0] = focal_x / z;
    j[1] = (R)0;
    j[2] = -(focal_x * x) / (z * z);
    j[3] = (R)0;
    j[4] = focal_y / z;
    j[5] = -(focal_y * y) / (z * z);
  }
}

void FN(orc_conic_from_J)(const R *sigma, const R *view, const R *J, int N, R mh_dist, R *conic, R *radius) {
  const R w00 = view[0], w01 = view[1], w02 = view[2], w10 = view[4], w11 = view[5], w12 = view[6], w20 = view[8],
          w21 = view[9], w22 = view[10];
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R *s = sigma + 6 * i, *j = J + 6 * i;
    const R s00 = s[0], s01 = s[1], s02 = s[2], s11 = s[3], s12 = s[4], s22 = s[5];
    const R j00 = j[0], j01 = j[1], j02 = j[2], j10 = j[3], j11 = j[4], j12 = j[5];
    const R m00 = j00 * w00 + j01 * w10 + j02 * w20, m01 = j00 * w01 + j01 * w11 + j02 * w21,
            m02 = j00 * w02 + j01 * w12 + j02 * w22;
    const R m10 = j10 * w00 + j11 * w10 + j12 * w20, m11 = j10 * w01 + j11 * w11 + j12 * w21,
            m12 = j10 * w02 + j11 * w12 + j12 * w22;
    const R v00 = s00 * m00 + s01 * m01 + s02 * m02, v01 = s00 * m10 + s01 * m11 + s02 * m12;
    const R v10 = s01 * m00 + s11 * m01 + s12 * m02, v11 = s01 * m10 + s11 * m11 + s12 * m12;
    const R v20 = s02 * m00 + s12 * m01 + s22 * m02, v21 = s02 * m10 + s12 * m11 + s22 * m12;
    const R cov00 = m00 * v00 + m01 * v10 + m02 * v20 + (R)0.3f;
    const R cov01 = m00 * v01 + m01 * v11 + m02 * v21;
    const R cov11 = m10 * v01 + m11 * v11 + m12 * v21 + (R)0.3f;
    const R det = cov00 * cov11 - cov01 * cov01;
    const R inv_det = (R)1 / det;
    conic[3 * i + 0] = cov11 * inv_det;
    conic[3 * i + 1] = -cov01 * inv_det;
    conic[3 * i + 2] = cov00 * inv_det;
    const R mid = (R)0.5 * (cov00 + cov11);
    const R lambda_term = SQRT(FMAX((R)0.1f, mid * mid - det));
    const R lambda1 = mid + lambda_term, lambda2 = mid - lambda_term;
    const R r_major = CEIL(mh_dist * SQRT(lambda1));
    const R r_minor = CEIL(mh_dist * SQRT(lambda2)); /* NaN when lambda2 < 0: kept (SURVEY 8a hazard 2) */
    const R th = (R)0.5 * ATAN2((R)2 * cov01, cov00 - cov11);
    radius[4 * i + 0] = r_major;
    radius[4 * i + 1] = r_minor;
    radius[4 * i + 2] = SIN(th);
    radius[4 * i + 3] = COS(th);
  }
}

void FN(orc_compute_conic)(const R *xyz, const R *view, const R *sigma, R focal_x, R focal_y, R tan_fovx, R tan_fovy,
                           R mh_dist, int N, R *J, R *conic, R *radius) {
  FN(orc_projection_jacobian)(xyz, focal_x, focal_y, tan_fovx, tan_fovy, N, J);
  FN(orc_conic_from_J)(sigma, view, J, N, mh_dist, conic, radius);
}

/* ------------------------------------------------------------------------------------ */
/* B1  get_sorted_gaussian_list   cuda/culling.cu:97-343, 386-475                        */
/* ------------------------------------------------------------------------------------ */

/* coarse candidate rectangle, cuda/culling.cu:209-224 */
static inline void FN(coarse_rect)(R u, R v, R r_major, int ntx, int nty, int *x0, int *x1, int *y0, int *y1) {
  const int radius_tiles = FN(f2i_sat)(CEIL(r_major * (R)0.0625f)) + 1;
  const int ptx = FN(f2i_sat)(FLOOR(u / (R)16));
  const int pty = FN(f2i_sat)(FLOOR(v / (R)16));
  /* long arithmetic so saturated values cannot overflow */
  long sx = (long)ptx - radius_tiles, ex = (long)ptx + radius_tiles + 1;
  long sy = (long)pty - radius_tiles, ey = (long)pty + radius_tiles + 1;
  if (sx < 0) sx = 0;
  if (ex > ntx) ex = ntx;
  if (sy < 0) sy = 0;
  if (ey > nty) ey = nty;
  if (ex < sx) ex = sx;
  if (ey < sy) ey = sy;
  *x0 = (int)sx; *x1 = (int)ex; *y0 = (int)sy; *y1 = (int)ey;
}

/* compute_obb, cuda/culling.cu:148-165 */
static inline void FN(compute_obb)(R u, R v, R r_major, R r_minor, R sin_t, R cos_t, R *obb) {
  const R v1x = r_major * cos_t, v1y = r_major * sin_t, v2x = -r_minor * sin_t, v2y = r_minor * cos_t;
  obb[0] = u - v1x - v2x; obb[1] = v - v1y - v2y;
  obb[2] = u + v1x - v2x; obb[3] = v + v1y - v2y;
  obb[4] = u - v1x + v2x; obb[5] = v - v1y + v2y;
  obb[6] = u + v1x + v2x; obb[7] = v + v1y + v2y;
}

/* split_axis_test, cuda/culling.cu:97-146.  fminf/fmaxf semantics: a NaN operand is
 * ignored, every comparison with NaN is false (so NaN corners can only pass). */
static inline int FN(sat_test)(const R *obb, const R *tb /* left,right,top,bottom */) {
  const R mnx = FMIN(FMIN(obb[0], obb[2]), FMIN(obb[4], obb[6]));
  const R mxx = FMAX(FMAX(obb[0], obb[2]), FMAX(obb[4], obb[6]));
  if (mnx > tb[1] || mxx < tb[0]) return 0;
  const R mny = FMIN(FMIN(obb[1], obb[3]), FMIN(obb[5], obb[7]));
  const R mxy = FMAX(FMAX(obb[1], obb[3]), FMAX(obb[5], obb[7]));
  if (mny > tb[3] || mxy < tb[2]) return 0;
  {
    const R ax = obb[2] - obb[0], ay = obb[3] - obb[1];
    const R tl = ax * tb[0] + ay * tb[2], tr = ax * tb[1] + ay * tb[2];
    const R bl = ax * tb[0] + ay * tb[3], br = ax * tb[1] + ay * tb[3];
    const R mn_t = FMIN(FMIN(tl, tr), FMIN(bl, br)), mx_t = FMAX(FMAX(tl, tr), FMAX(bl, br));
    const R pr = ax * obb[2] + ay * obb[3], pl = ax * obb[0] + ay * obb[1];
    const R mn_o = FMIN(pr, pl), mx_o = FMAX(pr, pl);
    if (mn_t > mx_o || mx_t < mn_o) return 0;
  }
  {
    const R ax = obb[2] - obb[6], ay = obb[3] - obb[7];
    const R tl = ax * tb[0] + ay * tb[2], tr = ax * tb[1] + ay * tb[2];
    const R bl = ax * tb[0] + ay * tb[3], br = ax * tb[1] + ay * tb[3];
    const R mn_t = FMIN(FMIN(tl, tr), FMIN(bl, br)), mx_t = FMAX(FMAX(tl, tr), FMAX(bl, br));
    const R pt = ax * obb[2] + ay * obb[3], pb = ax * obb[6] + ay * obb[7];
    const R mn_o = FMIN(pt, pb), mx_o = FMAX(pt, pb);
    if (mn_t > mx_o || mx_t < mn_o) return 0;
  }
  return 1;
}

/* Call 1 of the two-call protocol (sorted == nullptr): number of coarse candidate pairs,
 * cuda/culling.cu:226-232, :396-407. */
long FN(orc_count_tile_pairs)(const R *uv, const R *radius, int ntx, int nty, int N) {
  long total = 0;
  for (int i = 0; i < N; ++i) {
    int x0, x1, y0, y1;
    FN(coarse_rect)(uv[2 * i], uv[2 * i + 1], radius[4 * i], ntx, nty, &x0, &x1, &y0, &y1);
    total += (long)(x1 - x0) * (long)(y1 - y0);
  }
  return total;
}

typedef struct { int tile; R z; int id; } FN(splat_t);

static int FN(splat_cmp)(const void *a, const void *b) {
  const FN(splat_t) *p = (const FN(splat_t) *)a, *q = (const FN(splat_t) *)b;
  if (p->tile != q->tile) return p->tile < q->tile ? -1 : 1;
  if (p->z < q->z) return -1;
  if (p->z > q->z) return 1;
  return (p->id > q->id) - (p->id < q->id); /* reference: arbitrary on ties; we pin gaussian id */
}

/* Call 2: fills sorted[0..S) (gaussian ids ordered by (tile, z, id)) and ranges[0..T].
 * cuda/culling.cu:247-343, 409-475.  `sorted` must have room for orc_count_tile_pairs()
 * entries (as in the reference, where the buffer is sized by the candidate count).
 * Returns S, the number of (tile, gaussian) instances that pass the SAT test. */
long FN(orc_sorted_gaussian_list)(const R *uv, const R *xyz, const R *radius, int ntx, int nty, int N, int *sorted,
                                  int *ranges) {
  const int num_tiles = ntx * nty;
  long cap = FN(orc_count_tile_pairs)(uv, radius, ntx, nty, N);
  FN(splat_t) *sp = (FN(splat_t) *)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(FN(splat_t)));
  long S = 0;
  for (int i = 0; i < N; ++i) {
    int x0, x1, y0, y1;
    const R u = uv[2 * i], v = uv[2 * i + 1];
    FN(coarse_rect)(u, v, radius[4 * i], ntx, nty, &x0, &x1, &y0, &y1);
    if (x1 <= x0 || y1 <= y0) continue;
    R obb[8];
    FN(compute_obb)(u, v, radius[4 * i], radius[4 * i + 1], radius[4 * i + 2], radius[4 * i + 3], obb);
    for (int tx = x0; tx < x1; ++tx)
      for (int ty = y0; ty < y1; ++ty) {
        R tb[4];
        tb[0] = (R)tx * (R)16; tb[1] = (R)(tx + 1) * (R)16; tb[2] = (R)ty * (R)16; tb[3] = (R)(ty + 1) * (R)16;
        if (FN(sat_test)(obb, tb)) {
          sp[S].tile = ty * ntx + tx; sp[S].z = xyz[3 * i + 2]; sp[S].id = i;
          ++S;
        }
      }
  }
  qsort(sp, (size_t)S, sizeof(FN(splat_t)), FN(splat_cmp));
  for (long k = 0; k < S; ++k) sorted[k] = sp[k].id;
  /* find_tile_boundaries, cuda/culling.cu:302-343: ranges[t] = #instances with tile < t */
  {
    long k = 0;
    for (int t = 0; t <= num_tiles; ++t) {
      while (k < S && sp[k].tile < t) ++k;
      ranges[t] = (int)k;
    }
  }
  free(sp);
  return S;
}

/* ------------------------------------------------------------------------------------ */
/* S1  precompute_spherical_harmonics   cuda/spherical_harmonics.cu:8-94                 */
/* The basis comes from sphericart (un-vendored): real, orthonormal, no Condon-Shortley  */
/* phase, index l*l+l+m.  l<=2 pinned by tests/cuda_forward_test.cpp:541-627 and         */
/* tests/cuda_backward_test.cpp:610-624; all 16 functions and their gradients pinned      */
/* against scipy.special.sph_harm_y (tests/test_oracle_known_answers.py, r04).           */
/* ------------------------------------------------------------------------------------ */
#define SH_C0 ((R)0.28209479177387814)
#define SH_C1 ((R)0.4886025119029199)
#define SH_C2 ((R)1.0925484305920792)
#define SH_C3 ((R)0.31539156525252005)
#define SH_C4 ((R)0.5462742152960396)
#define SH_C5 ((R)0.5900435899266435)
#define SH_C6 ((R)2.890611442640554)
#define SH_C7 ((R)0.4570457994644658)
#define SH_C8 ((R)0.3731763325901154)
#define SH_C9 ((R)1.445305721320277)

/* Homogeneous-polynomial (solid harmonic) form; equals Y_lm on the unit sphere. */
static inline void FN(sh_basis)(int l_max, R x, R y, R z, R *Y) {
  Y[0] = SH_C0;
  if (l_max < 1) return;
  Y[1] = SH_C1 * y; Y[2] = SH_C1 * z; Y[3] = SH_C1 * x;
  if (l_max < 2) return;
  const R xx = x * x, yy = y * y, zz = z * z;
  Y[4] = SH_C2 * (x * y);
  Y[5] = SH_C2 * (y * z);
  Y[6] = SH_C3 * ((R)2 * zz - xx - yy);
  Y[7] = SH_C2 * (x * z);
  Y[8] = SH_C4 * (xx - yy);
  if (l_max < 3) return;
  Y[9] = SH_C5 * (y * ((R)3 * xx - yy));
  Y[10] = SH_C6 * (x * y * z);
  Y[11] = SH_C7 * (y * ((R)4 * zz - xx - yy));
  Y[12] = SH_C8 * (z * ((R)2 * zz - (R)3 * xx - (R)3 * yy));
  Y[13] = SH_C7 * (x * ((R)4 * zz - xx - yy));
  Y[14] = SH_C9 * (z * (xx - yy));
  Y[15] = SH_C5 * (x * (xx - (R)3 * yy));
}

/* Cartesian gradient of the same polynomials, dY[k][axis]. */
static inline void FN(sh_basis_grad)(int l_max, R x, R y, R z, R (*dY)[3]) {
  dY[0][0] = dY[0][1] = dY[0][2] = (R)0;
  if (l_max < 1) return;
  dY[1][0] = 0; dY[1][1] = SH_C1; dY[1][2] = 0;
  dY[2][0] = 0; dY[2][1] = 0; dY[2][2] = SH_C1;
  dY[3][0] = SH_C1; dY[3][1] = 0; dY[3][2] = 0;
  if (l_max < 2) return;
  const R xx = x * x, yy = y * y, zz = z * z;
  dY[4][0] = SH_C2 * y; dY[4][1] = SH_C2 * x; dY[4][2] = 0;
  dY[5][0] = 0; dY[5][1] = SH_C2 * z; dY[5][2] = SH_C2 * y;
  dY[6][0] = SH_C3 * ((R)-2 * x); dY[6][1] = SH_C3 * ((R)-2 * y); dY[6][2] = SH_C3 * ((R)4 * z);
  dY[7][0] = SH_C2 * z; dY[7][1] = 0; dY[7][2] = SH_C2 * x;
  dY[8][0] = SH_C4 * ((R)2 * x); dY[8][1] = SH_C4 * ((R)-2 * y); dY[8][2] = 0;
  if (l_max < 3) return;
  dY[9][0] = SH_C5 * ((R)6 * x * y); dY[9][1] = SH_C5 * ((R)3 * xx - (R)3 * yy); dY[9][2] = 0;
  dY[10][0] = SH_C6 * (y * z); dY[10][1] = SH_C6 * (x * z); dY[10][2] = SH_C6 * (x * y);
  dY[11][0] = SH_C7 * ((R)-2 * x * y); dY[11][1] = SH_C7 * ((R)4 * zz - xx - (R)3 * yy); dY[11][2] = SH_C7 * ((R)8 * y * z);
  dY[12][0] = SH_C8 * ((R)-6 * x * z); dY[12][1] = SH_C8 * ((R)-6 * y * z);
  dY[12][2] = SH_C8 * ((R)6 * zz - (R)3 * xx - (R)3 * yy);
  dY[13][0] = SH_C7 * ((R)4 * zz - (R)3 * xx - yy); dY[13][1] = SH_C7 * ((R)-2 * x * y); dY[13][2] = SH_C7 * ((R)8 * x * z);
  dY[14][0] = SH_C9 * ((R)2 * x * z); dY[14][1] = SH_C9 * ((R)-2 * y * z); dY[14][2] = SH_C9 * (xx - yy);
  dY[15][0] = SH_C5 * ((R)3 * xx - (R)3 * yy); dY[15][1] = SH_C5 * ((R)-6 * x * y); dY[15][2] = 0;
}

void FN(orc_sh_forward)(const R *xyz, const R *sh, const R *band0, const R *campos, int l_max, int N, R *rgb) {
  const int n = (l_max + 1) * (l_max + 1);
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    /* compute_dir_kernel :8-26 */
    const R dx = xyz[3 * i] - campos[0], dy = xyz[3 * i + 1] - campos[1], dz = xyz[3 * i + 2] - campos[2];
    const R len = SQRT(dx * dx + dy * dy + dz * dz) + (R)1e-9f;
    R Y[16];
    FN(sh_basis)(l_max, dx / len, dy / len, dz / len, Y);
    /* compute_rgb_from_sh_kernel :28-60 (no clamp, no sigmoid) */
    R r = band0[3 * i + 0] * Y[0] + (R)0.5, g = band0[3 * i + 1] * Y[0] + (R)0.5, b = band0[3 * i + 2] * Y[0] + (R)0.5;
    if (n > 1) {
      const R *c = sh + (size_t)i * (n - 1) * 3;
      for (int k = 0; k < n - 1; ++k) {
        r += c[3 * k + 0] * Y[k + 1];
        g += c[3 * k + 1] * Y[k + 1];
        b += c[3 * k + 2] * Y[k + 1];
      }
    }
    rgb[3 * i + 0] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
  }
}

/* ------------------------------------------------------------------------------------ */
/* R1  render_image   cuda/render.cu:6-135                                               */
/* Per-pixel restatement.  The warp-level loop exit (`any_active`) only stops work once   */
/* every pixel of the tile is done, and done pixels are frozen, so every output is a      */
/* per-pixel property: n = (index of the splat that trips T<1e-4)+1, else the list length. */
/* The row polynomial basic + linear*i + quad*i*i is kept (lane base row = 8*(py/8)).     */
/* ------------------------------------------------------------------------------------ */
void FN(orc_render_image)(const R *uv, const R *opacity, const R *conic, const R *rgb, R bg, const int *sorted,
                          const int *ranges, int width, int height, int *n_out, R *T_out, R *image, int threads) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16;
  const int num_tiles = ntx * nty;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
  for (int tile = 0; tile < num_tiles; ++tile) {
    const int start = ranges[tile], end = ranges[tile + 1], total = end - start;
    const int tx = tile % ntx, ty = tile / ntx;
    for (int ly = 0; ly < 16; ++ly)
      for (int lx = 0; lx < 16; ++lx) {
        const int px = tx * 16 + lx, py = ty * 16 + ly;
        if (px >= width || py >= height) continue;
        const int base_row = ty * 16 + (ly / 8) * 8; /* lane's first row */
        const int i = ly % 8;
        R T = (R)1, ar = 0, ag = 0, ab = 0;
        int n = 0, done = 0;
        for (int k = 0; k < total; ++k) {
          if (done) break; /* done pixels no longer change; the warp loop may run on */
          const int g = sorted[start + k];
          const R opa = (R)1 / ((R)1 + EXP(-opacity[g]));
          const R dx = uv[2 * g] - (R)px, dy = uv[2 * g + 1] - (R)base_row;
          const R a = conic[3 * g], b = conic[3 * g + 1], c = conic[3 * g + 2];
          const R basic = (R)-0.5 * (a * dx * dx + (R)2 * b * dx * dy + c * dy * dy);
          const R linear = c * dy + b * dx;
          const R quad = (R)-0.5 * c;
          n += 1; /* num_splats += !done, before the update (:70) */
          const R power = FMIN((R)0, basic + linear * (R)i + quad * (R)i * (R)i);
          R alpha = FMIN((R)0.99f, opa * EXP(power));
          alpha = (alpha > (R)0.00392156862f) ? alpha : (R)0;
          const R test_T = T * ((R)1 - alpha);
          done = test_T < (R)0.0001f;
          const R w = alpha * T;
          ar += rgb[3 * g] * w; ag += rgb[3 * g + 1] * w; ab += rgb[3 * g + 2] * w;
          T = test_T; /* the splat that trips `done` is still accumulated (:81-87) */
        }
        const int pid = py * width + px;
        n_out[pid] = n;
        T_out[pid] = T;
        image[3 * pid + 0] = ar + T * bg;
        image[3 * pid + 1] = ag + T * bg;
        image[3 * pid + 2] = ab + T * bg;
      }
  }
}

/* S_eff = sum over tiles of max over the tile's pixels of n. */
long FN(orc_effective_instances)(const int *n, int width, int height) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16;
  long total = 0;
  for (int ty = 0; ty < nty; ++ty)
    for (int tx = 0; tx < ntx; ++tx) {
      int m = 0;
      for (int ly = 0; ly < 16; ++ly)
        for (int lx = 0; lx < 16; ++lx) {
          const int px = tx * 16 + lx, py = ty * 16 + ly;
          if (px < width && py < height && n[py * width + px] > m) m = n[py * width + px];
        }
      total += m;
    }
  return total;
}

/* ------------------------------------------------------------------------------------ */
/* R2  render_image_backward   cuda/render_backward.cu:11-258                            */
/* Keeps the reference's work decomposition: one 32-lane warp per tile, lane -> column    */
/* lane%16, rows (lane/16)*8 .. +7; per-lane partial sums, then a 32-lane reduction, the   */
/* `any(grad_opacity != 0)` gate (:170) and the x0.5*W / x0.5*H on the uv gradient         */
/* (:186-187).  Outputs are accumulated (+=): callers pre-zero them.                       */
/* Phase 1 (parallel over tiles) writes one 9-vector per list entry, phase 2 adds them in   */
/* list order, so the result does not depend on the thread count.                         */
/* ------------------------------------------------------------------------------------ */
void FN(orc_render_image_backward)(const R *uv, const R *opacity, const R *conic, const R *rgb, R bg,
                                   const int *sorted, const int *ranges, const int *n_px, const R *T_px,
                                   const R *grad_image, int width, int height, R *grad_rgb, R *grad_opacity,
                                   R *grad_uv, R *grad_conic, int threads) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16;
  const int num_tiles = ntx * nty;
  const long S = ranges[num_tiles];
  R *part = (R *)calloc((size_t)(S > 0 ? S : 1) * 9, sizeof(R));
  unsigned char *have = (unsigned char *)calloc((size_t)(S > 0 ? S : 1), 1);
  (void)threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
  for (int tile = 0; tile < num_tiles; ++tile) {
    const int start = ranges[tile];
    const int tx = tile % ntx, ty = tile / ntx;
    /* per-pixel state, [lane][i] */
    R T[32][8], Tf[32][8], acc[32][8][3], gi[32][8][3];
    int npx[32][8], valid[32][8];
    int top = 0;
    for (int lane = 0; lane < 32; ++lane)
      for (int i = 0; i < 8; ++i) {
        const int px = tx * 16 + lane % 16, py = ty * 16 + (lane / 16) * 8 + i;
        const int ok = px < width && py < height;
        valid[lane][i] = ok;
        if (ok) {
          const int pid = py * width + px;
          npx[lane][i] = n_px[pid];
          T[lane][i] = T_px[pid];
          gi[lane][i][0] = grad_image[3 * pid]; gi[lane][i][1] = grad_image[3 * pid + 1];
          gi[lane][i][2] = grad_image[3 * pid + 2];
          if (npx[lane][i] > top) top = npx[lane][i];
        } else {
          npx[lane][i] = 0; T[lane][i] = 0; gi[lane][i][0] = gi[lane][i][1] = gi[lane][i][2] = 0;
        }
        Tf[lane][i] = T[lane][i];
        acc[lane][i][0] = acc[lane][i][1] = acc[lane][i][2] = 0;
      }
    for (int idx = top - 1; idx >= 0; --idx) {
      const int g = sorted[start + idx];
      const R a = conic[3 * g], b = conic[3 * g + 1], c = conic[3 * g + 2];
      const R cr = rgb[3 * g], cg_ = rgb[3 * g + 1], cb = rgb[3 * g + 2];
      const R opa = (R)1 / ((R)1 + EXP(-opacity[g]));
      R red[32][9];
      int any_nz = 0;
      for (int lane = 0; lane < 32; ++lane) {
        const int bx = tx * 16 + lane % 16, by = ty * 16 + (lane / 16) * 8;
        const R dx = uv[2 * g] - (R)bx, dy = uv[2 * g + 1] - (R)by;
        const R basic = (R)-0.5 * (a * dx * dx + (R)2 * b * dx * dy + c * dy * dy);
        const R linear = c * dy + b * dx;
        const R quad = (R)-0.5 * c;
        R g_r = 0, g_g = 0, g_b = 0, g_o = 0, g_basic = 0, g_lin = 0, g_quad = 0;
        for (int i = 0; i < 8; ++i) {
          const R power = FMIN((R)0, basic + linear * (R)i + quad * (R)i * (R)i);
          R gg = EXP(power);
          R alpha = FMIN((R)0.99f, opa * gg);
          const int vs = valid[lane][i] && (alpha >= (R)0.00392156862f) && (idx < npx[lane][i]);
          /* `if (valid_mask)` (:128) only skips work that is a no-op for !vs lanes */
          if (!vs) continue;
          T[lane][i] *= (R)1 / ((R)1 - alpha);
          const R Ti = T[lane][i];
          g_r += alpha * Ti * gi[lane][i][0];
          g_g += alpha * Ti * gi[lane][i][1];
          g_b += alpha * Ti * gi[lane][i][2];
          R ga = 0;
          ga += (cr - acc[lane][i][0]) * gi[lane][i][0];
          ga += (cg_ - acc[lane][i][1]) * gi[lane][i][1];
          ga += (cb - acc[lane][i][2]) * gi[lane][i][2];
          ga *= Ti;
          R bgdot = 0;
          bgdot += bg * gi[lane][i][0]; bgdot += bg * gi[lane][i][1]; bgdot += bg * gi[lane][i][2];
          ga += (-Tf[lane][i] / ((R)1 - alpha)) * bgdot;
          g_o += gg * ga * opa * ((R)1 - opa);
          acc[lane][i][0] = alpha * cr + ((R)1 - alpha) * acc[lane][i][0];
          acc[lane][i][1] = alpha * cg_ + ((R)1 - alpha) * acc[lane][i][1];
          acc[lane][i][2] = alpha * cb + ((R)1 - alpha) * acc[lane][i][2];
          const R gpow = gg * (ga * opa);
          g_basic += gpow; g_lin += gpow * (R)i; g_quad += gpow * (R)i * (R)i;
        }
        if (g_o != (R)0) any_nz = 1;
        red[lane][0] = g_r; red[lane][1] = g_g; red[lane][2] = g_b; red[lane][3] = g_o;
        /* conic (:192-196) */
        red[lane][4] = g_basic * ((R)-0.5 * dx * dx);
        red[lane][5] = g_basic * (-dx * dy) + g_lin * dx;
        red[lane][6] = g_basic * ((R)-0.5 * dy * dy) + (g_lin * dy) - ((R)0.5 * g_quad);
        /* uv (:180-187) */
        red[lane][7] = ((-a * dx - b * dy) * g_basic + b * g_lin) * ((R)0.5 * (R)width);
        red[lane][8] = ((-c * dy - b * dx) * g_basic + c * g_lin) * ((R)0.5 * (R)height);
      }
      if (!any_nz) continue; /* :170 -- skips ALL nine outputs */
      for (int off = 16; off > 0; off >>= 1)
        for (int lane = 0; lane < off; ++lane)
          for (int k = 0; k < 9; ++k) red[lane][k] += red[lane + off][k];
      R *p = part + (size_t)(start + idx) * 9;
      for (int k = 0; k < 9; ++k) p[k] = red[0][k];
      have[start + idx] = 1;
    }
  }
  for (long s = 0; s < S; ++s) {
    if (!have[s]) continue;
    const int g = sorted[s];
    const R *p = part + (size_t)s * 9;
    grad_rgb[3 * g] += p[0]; grad_rgb[3 * g + 1] += p[1]; grad_rgb[3 * g + 2] += p[2];
    grad_opacity[g] += p[3];
    grad_conic[3 * g] += p[4]; grad_conic[3 * g + 1] += p[5]; grad_conic[3 * g + 2] += p[6];
    grad_uv[2 * g] += p[7]; grad_uv[2 * g + 1] += p[8];
  }
  free(part);
  free(have);
}

/* ------------------------------------------------------------------------------------ */
/* Q1  project_to_screen_backward   cuda/projection_backward.cu:6-75   (+=)              */
/* ------------------------------------------------------------------------------------ */
void FN(orc_project_to_screen_backward)(const R *xyz_c, const R *proj, const R *uv_grad, int N, int width, int height,
                                        R *xyz_c_grad) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R x = xyz_c[3 * i], y = xyz_c[3 * i + 1], z = xyz_c[3 * i + 2];
    const R x_clip = proj[0] * x + proj[1] * y + proj[2] * z + proj[3];
    const R y_clip = proj[4] * x + proj[5] * y + proj[6] * z + proj[7];
    const R w_clip = proj[12] * x + proj[13] * y + proj[14] * z + proj[15];
    if (FABS(w_clip) < (R)1e-6f) continue;
    const R w_inv = (R)1 / w_clip, w_inv2 = w_inv * w_inv;
    const R dx_ndc = uv_grad[2 * i] * (R)width * (R)0.5, dy_ndc = uv_grad[2 * i + 1] * (R)height * (R)0.5;
    const R dx_clip = dx_ndc * w_inv, dy_clip = dy_ndc * w_inv;
    const R dw_clip = -dx_ndc * x_clip * w_inv2 - dy_ndc * y_clip * w_inv2;
    const R dz_clip = (R)0;
    xyz_c_grad[3 * i + 0] += proj[0] * dx_clip + proj[4] * dy_clip + proj[8] * dz_clip + proj[12] * dw_clip;
    xyz_c_grad[3 * i + 1] += proj[1] * dx_clip + proj[5] * dy_clip + proj[9] * dz_clip + proj[13] * dw_clip;
    xyz_c_grad[3 * i + 2] += proj[2] * dx_clip + proj[6] * dy_clip + proj[10] * dz_clip + proj[14] * dw_clip;
  }
}

/* Q2  compute_camera_space_points_backward   cuda/projection_backward.cu:95-137   (+=) */
void FN(orc_camera_space_points_backward)(const R *xyz_w, const R *view, const R *xyz_c_grad, int N, R *xyz_w_grad) {
  (void)xyz_w;
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R gx = xyz_c_grad[3 * i], gy = xyz_c_grad[3 * i + 1], gz = xyz_c_grad[3 * i + 2];
    xyz_w_grad[3 * i + 0] += view[0] * gx + view[4] * gy + view[8] * gz;
    xyz_w_grad[3 * i + 1] += view[1] * gx + view[5] * gy + view[9] * gz;
    xyz_w_grad[3 * i + 2] += view[2] * gx + view[6] * gy + view[10] * gz;
  }
}

/* H1  compute_projection_jacobian_backward   cuda/gaussian_backward.cu:6-78   (+=) */
void FN(orc_projection_jacobian_backward)(const R *xyz, R focal_x, R focal_y, R tan_fovx, R tan_fovy, const R *J_grad,
                                          int N, R *xyz_grad) {
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (FABS(z) < (R)1e-6f) continue;
    const R z_inv = (R)1 / (z + (R)1e-6f), z_inv2 = z_inv * z_inv, z_inv3 = z_inv2 * z_inv;
    const R limx = (R)1.3f * tan_fovx, limy = (R)1.3f * tan_fovy;
    const R txtz = x * z_inv, tytz = y * z_inv;
    const R dJ00 = J_grad[6 * i + 0], dJ02 = J_grad[6 * i + 2], dJ11 = J_grad[6 * i + 4], dJ12 = J_grad[6 * i + 5];
    R dx = 0, dy = 0, dz = 0;
    dz += dJ00 * (-focal_x * z_inv2);
    if (FABS(txtz) <= limx) {
      dx += dJ02 * (-focal_x * z_inv2);
      dz += dJ02 * ((R)2 * focal_x * x * z_inv3);
    } else {
      const R cx = (txtz > (R)0 ? limx : -limx);
      dz += dJ02 * (focal_x * cx * z_inv2);
    }
    dz += dJ11 * (-focal_y * z_inv2);
    if (FABS(tytz) <= limy) {
      dy += dJ12 * (-focal_y * z_inv2);
      dz += dJ12 * ((R)2 * focal_y * y * z_inv3);
    } else {
      const R cy = (tytz > (R)0 ? limy : -limy);
      dz += dJ12 * (focal_y * cy * z_inv2);
    }
    xyz_grad[3 * i] += dx; xyz_grad[3 * i + 1] += dy; xyz_grad[3 * i + 2] += dz;
  }
}

/* H2  compute_conic_backward   cuda/gaussian_backward.cu:97-248   (+= J_grad, sigma_grad) */
void FN(orc_conic_backward)(const R *J, const R *sigma, const R *view, const R *conic, const R *conic_grad, int N,
                            R *J_grad, R *sigma_grad) {
  const R w00 = view[0], w01 = view[1], w02 = view[2], w10 = view[4], w11 = view[5], w12 = view[6], w20 = view[8],
          w21 = view[9], w22 = view[10];
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R *j = J + 6 * i, *s = sigma + 6 * i;
    const R j00 = j[0], j01 = j[1], j02 = j[2], j10 = j[3], j11 = j[4], j12 = j[5];
    const R s00 = s[0], s01 = s[1], s02 = s[2], s11 = s[3], s12 = s[4], s22 = s[5];
    const R m00 = j00 * w00 + j01 * w10 + j02 * w20, m01 = j00 * w01 + j01 * w11 + j02 * w21,
            m02 = j00 * w02 + j01 * w12 + j02 * w22;
    const R m10 = j10 * w00 + j11 * w10 + j12 * w20, m11 = j10 * w01 + j11 * w11 + j12 * w21,
            m12 = j10 * w02 + j11 * w12 + j12 * w22;
    const R v00 = s00 * m00 + s01 * m01 + s02 * m02, v01 = s00 * m10 + s01 * m11 + s02 * m12;
    const R v10 = s01 * m00 + s11 * m01 + s12 * m02, v11 = s01 * m10 + s11 * m11 + s12 * m12;
    const R v20 = s02 * m00 + s12 * m01 + s22 * m02, v21 = s02 * m10 + s12 * m11 + s22 * m12;
    const R dc00 = conic_grad[3 * i], dc01 = conic_grad[3 * i + 1], dc11 = conic_grad[3 * i + 2];
    const R c00 = conic[3 * i], c01 = conic[3 * i + 1], c11 = conic[3 * i + 2];
    const R t00 = c00 * dc00 + c01 * dc01, t01 = c00 * dc01 + c01 * dc11;
    const R t10 = c01 * dc00 + c11 * dc01, t11 = c01 * dc01 + c11 * dc11;
    const R d_c00 = -(t00 * c00 + t01 * c01);
    const R d_c01 = -(t00 * c01 + t01 * c11);
    const R d_c11 = -(t10 * c01 + t11 * c11);
    const R dv00 = d_c00 * m00 + d_c01 * m10, dv01 = d_c01 * m00 + d_c11 * m10;
    const R dv10 = d_c00 * m01 + d_c01 * m11, dv11 = d_c01 * m01 + d_c11 * m11;
    const R dv20 = d_c00 * m02 + d_c01 * m12, dv21 = d_c01 * m02 + d_c11 * m12;
    R *sg = sigma_grad + 6 * i;
    sg[0] += dv00 * m00 + dv01 * m10;
    sg[1] += dv00 * m01 + dv01 * m11 + dv10 * m00 + dv11 * m10;
    sg[2] += dv00 * m02 + dv01 * m12 + dv20 * m00 + dv21 * m10;
    sg[3] += dv10 * m01 + dv11 * m11;
    sg[4] += dv10 * m02 + dv11 * m12 + dv20 * m01 + dv21 * m11;
    sg[5] += dv20 * m02 + dv21 * m12;
    const R dmc00 = d_c00 * v00 + d_c01 * v01, dmc01 = d_c00 * v10 + d_c01 * v11, dmc02 = d_c00 * v20 + d_c01 * v21;
    const R dmc10 = d_c01 * v00 + d_c11 * v01, dmc11 = d_c01 * v10 + d_c11 * v11, dmc12 = d_c01 * v20 + d_c11 * v21;
    const R dmv00 = dv00 * s00 + dv10 * s01 + dv20 * s02, dmv01 = dv00 * s01 + dv10 * s11 + dv20 * s12,
            dmv02 = dv00 * s02 + dv10 * s12 + dv20 * s22;
    const R dmv10 = dv01 * s00 + dv11 * s01 + dv21 * s02, dmv11 = dv01 * s01 + dv11 * s11 + dv21 * s12,
            dmv12 = dv01 * s02 + dv11 * s12 + dv21 * s22;
    const R dm00 = dmc00 + dmv00, dm01 = dmc01 + dmv01, dm02 = dmc02 + dmv02;
    const R dm10 = dmc10 + dmv10, dm11 = dmc11 + dmv11, dm12 = dmc12 + dmv12;
    R *jg = J_grad + 6 * i;
    jg[0] += dm00 * w00 + dm01 * w01 + dm02 * w02;
    jg[1] += dm00 * w10 + dm01 * w11 + dm02 * w12;
    jg[2] += dm00 * w20 + dm01 * w21 + dm02 * w22;
    jg[3] += dm10 * w00 + dm11 * w01 + dm12 * w02;
    jg[4] += dm10 * w10 + dm11 * w11 + dm12 * w12;
    jg[5] += dm10 * w20 + dm11 * w21 + dm12 * w22;
  }
}

/* H3  compute_sigma_backward   cuda/gaussian_backward.cu:271-415   (= overwrite) */
void FN(orc_sigma_backward)(const R *q, const R *s, const R *dSigma_in, int N, R *dQ, R *dS) {
  for (int idx = 0; idx < N; ++idx) {
    const R qw = q[4 * idx], qx = q[4 * idx + 1], qy = q[4 * idx + 2], qz = q[4 * idx + 3];
    const R norm = SQRT(qw * qw + qx * qx + qy * qy + qz * qz);
    const R inv_norm = (R)1 / (norm + (R)1e-6f);
    const R w = qw * inv_norm, x = qx * inv_norm, y = qy * inv_norm, z = qz * inv_norm;
    const R Sx = EXP(s[3 * idx]), Sy = EXP(s[3 * idx + 1]), Sz = EXP(s[3 * idx + 2]);
    R Rm[9], M[9], dSg[9], dM[9], dR[9];
    Rm[0] = (R)1 - (R)2 * (y * y + z * z); Rm[1] = (R)2 * (x * y - w * z); Rm[2] = (R)2 * (x * z + w * y);
    Rm[3] = (R)2 * (x * y + w * z); Rm[4] = (R)1 - (R)2 * (x * x + z * z); Rm[5] = (R)2 * (y * z - w * x);
    Rm[6] = (R)2 * (x * z - w * y); Rm[7] = (R)2 * (y * z + w * x); Rm[8] = (R)1 - (R)2 * (x * x + y * y);
    M[0] = Rm[0] * Sx; M[1] = Rm[1] * Sy; M[2] = Rm[2] * Sz;
    M[3] = Rm[3] * Sx; M[4] = Rm[4] * Sy; M[5] = Rm[5] * Sz;
    M[6] = Rm[6] * Sx; M[7] = Rm[7] * Sy; M[8] = Rm[8] * Sz;
    const R *g = dSigma_in + 6 * idx;
    dSg[0] = g[0]; dSg[1] = (R)0.5 * g[1]; dSg[2] = (R)0.5 * g[2];
    dSg[3] = (R)0.5 * g[1]; dSg[4] = g[3]; dSg[5] = (R)0.5 * g[4];
    dSg[6] = (R)0.5 * g[2]; dSg[7] = (R)0.5 * g[4]; dSg[8] = g[5];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        dM[3 * r + c] = (R)2 * (dSg[3 * r] * M[c] + dSg[3 * r + 1] * M[3 + c] + dSg[3 * r + 2] * M[6 + c]);
    dR[0] = dM[0] * Sx; dR[1] = dM[1] * Sy; dR[2] = dM[2] * Sz;
    dR[3] = dM[3] * Sx; dR[4] = dM[4] * Sy; dR[5] = dM[5] * Sz;
    dR[6] = dM[6] * Sx; dR[7] = dM[7] * Sy; dR[8] = dM[8] * Sz;
    const R dsx = Rm[0] * dM[0] + Rm[3] * dM[3] + Rm[6] * dM[6];
    const R dsy = Rm[1] * dM[1] + Rm[4] * dM[4] + Rm[7] * dM[7];
    const R dsz = Rm[2] * dM[2] + Rm[5] * dM[5] + Rm[8] * dM[8];
    dS[3 * idx] = dsx * Sx; dS[3 * idx + 1] = dsy * Sy; dS[3 * idx + 2] = dsz * Sz;
    R dw = 0, dx = 0, dy = 0, dz = 0;
    dw += dR[1] * ((R)-2 * z) + dR[2] * ((R)2 * y);
    dx += dR[1] * ((R)2 * y) + dR[2] * ((R)2 * z);
    dy += dR[0] * ((R)-4 * y) + dR[1] * ((R)2 * x) + dR[2] * ((R)2 * w);
    dz += dR[0] * ((R)-4 * z) + dR[1] * ((R)-2 * w) + dR[2] * ((R)2 * x);
    dw += dR[3] * ((R)2 * z) + dR[5] * ((R)-2 * x);
    dx += dR[3] * ((R)2 * y) + dR[4] * ((R)-4 * x) + dR[5] * ((R)-2 * w);
    dy += dR[3] * ((R)2 * x) + dR[5] * ((R)2 * z);
    dz += dR[3] * ((R)2 * w) + dR[4] * ((R)-4 * z) + dR[5] * ((R)2 * y);
    dw += dR[6] * ((R)-2 * y) + dR[7] * ((R)2 * x);
    dx += dR[6] * ((R)2 * z) + dR[7] * ((R)2 * w) + dR[8] * ((R)-4 * x);
    dy += dR[6] * ((R)-2 * w) + dR[7] * ((R)2 * z) + dR[8] * ((R)-4 * y);
    dz += dR[6] * ((R)2 * x) + dR[7] * ((R)2 * y);
    const R dot = w * dw + x * dx + y * dy + z * dz;
    dQ[4 * idx + 0] = inv_norm * (dw - dot * w);
    dQ[4 * idx + 1] = inv_norm * (dx - dot * x);
    dQ[4 * idx + 2] = inv_norm * (dy - dot * y);
    dQ[4 * idx + 3] = inv_norm * (dz - dot * z);
  }
}

/* S2  precompute_spherical_harmonics_backward   cuda/spherical_harmonics_backward.cu:8-166 */
/* sh_grad (=), band0_grad (=), xyz_grad (+=).  dsph is taken as the Cartesian gradient of    */
/* the basis, layout [coeff][axis] (:74-76,108-110); the trailing (I - d d^T)/dist makes the    */
/* result independent of how the basis is extended off the unit sphere.                       */
void FN(orc_sh_backward)(const R *xyz, const R *band0, const R *sh, const R *campos, const R *rgb_grad, int l_max, int N,
                         R *sh_grad, R *band0_grad, R *xyz_grad) {
  const int n = (l_max + 1) * (l_max + 1);
#pragma omp parallel for schedule(static) num_threads(orc_threads)  /* independent per gaussian: same arithmetic at any thread count */
  for (int i = 0; i < N; ++i) {
    const R fx = xyz[3 * i] - campos[0], fy = xyz[3 * i + 1] - campos[1], fz = xyz[3 * i + 2] - campos[2];
    const R len = SQRT(fx * fx + fy * fy + fz * fz) + (R)1e-9f;
    const R ux = fx / len, uy = fy / len, uz = fz / len;
    R Y[16], dY[16][3];
    FN(sh_basis)(l_max, ux, uy, uz, Y);
    FN(sh_basis_grad)(l_max, ux, uy, uz, dY);
    const R *gr = rgb_grad + 3 * i;
    band0_grad[3 * i] = gr[0] * Y[0]; band0_grad[3 * i + 1] = gr[1] * Y[0]; band0_grad[3 * i + 2] = gr[2] * Y[0];
    R dRx = 0, dGx = 0, dBx = 0, dRy = 0, dGy = 0, dBy = 0, dRz = 0, dGz = 0, dBz = 0;
    const R R0 = band0[3 * i], G0 = band0[3 * i + 1], B0 = band0[3 * i + 2];
    dRx += dY[0][0] * R0; dGx += dY[0][0] * G0; dBx += dY[0][0] * B0;
    dRy += dY[0][1] * R0; dGy += dY[0][1] * G0; dBy += dY[0][1] * B0;
    dRz += dY[0][2] * R0; dGz += dY[0][2] * G0; dBz += dY[0][2] * B0;
    if (n > 1) {
      R *sg = sh_grad + (size_t)i * (n - 1) * 3;
      const R *c = sh + (size_t)i * (n - 1) * 3;
      for (int k = 0; k < n - 1; ++k) {
        const R yv = Y[k + 1];
        sg[3 * k] = gr[0] * yv; sg[3 * k + 1] = gr[1] * yv; sg[3 * k + 2] = gr[2] * yv;
        const R ddx = dY[k + 1][0], ddy = dY[k + 1][1], ddz = dY[k + 1][2];
        const R Ri = c[3 * k], Gi = c[3 * k + 1], Bi = c[3 * k + 2];
        dRx += ddx * Ri; dGx += ddx * Gi; dBx += ddx * Bi;
        dRy += ddy * Ri; dGy += ddy * Gi; dBy += ddy * Bi;
        dRz += ddz * Ri; dGz += ddz * Gi; dBz += ddz * Bi;
      }
    }
    const R tx = gr[0] * dRx + gr[1] * dGx + gr[2] * dBx;
    const R ty = gr[0] * dRy + gr[1] * dGy + gr[2] * dBy;
    const R tz = gr[0] * dRz + gr[1] * dGz + gr[2] * dBz;
    const R dist = SQRT(fx * fx + fy * fy + fz * fz) + (R)1e-9f;
    const R dxn = fx / dist, dyn = fy / dist, dzn = fz / dist;
    const R dot = tx * dxn + ty * dyn + tz * dzn;
    xyz_grad[3 * i] += (tx - dot * dxn) / dist;
    xyz_grad[3 * i + 1] += (ty - dot * dyn) / dist;
    xyz_grad[3 * i + 2] += (tz - dot * dzn) / dist;
  }
}

/* ------------------------------------------------------------------------------------ */
/* C1  compact_masked_array / scatter_masked_array   include/gsplat_cuda/cuda_data.cuh:106-167 */
/* ------------------------------------------------------------------------------------ */
int FN(orc_compact_masked)(const R *src, const unsigned char *mask, int N, int stride, R *dst) {
  int m = 0;
  for (int i = 0; i < N; ++i)
    if (mask[i]) {
      for (int k = 0; k < stride; ++k) dst[(size_t)m * stride + k] = src[(size_t)i * stride + k];
      ++m;
    }
  return m;
}

void FN(orc_scatter_masked)(const R *src, const unsigned char *mask, int N, int stride, R *dst) {
  int m = 0;
  for (int i = 0; i < N; ++i)
    if (mask[i]) {
      for (int k = 0; k < stride; ++k) dst[(size_t)i * stride + k] = src[(size_t)m * stride + k];
      ++m;
    }
}

/* ------------------------------------------------------------------------------------ */
/* F1  fused_loss / compute_psnr   cuda/loss.cu:58-471, 476-525   ("next" row f1)          */
/* Loss = mean over H*W*3 of (1-w)|p-g| + w(1-SSIM), SSIM from an 11-tap separable Gaussian */
/* window with CLAMPED borders on pred/gt (:41-46); the gradient convolves the three partial */
/* derivative maps with ZERO padding (:48-53, 333-337), adds (1-w)*(p>g ? 1 : -1) and scales  */
/* by 1/(H*W*3) (:419-428).  Interleaved RGB.                                              */
/* ------------------------------------------------------------------------------------ */
static const double FN(kGauss11)[11] = {0.001028380123898387,  0.0075987582094967365, 0.036000773310661316,
                                        0.10936068743467331,   0.21300552785396576,   0.26601171493530273,
                                        0.21300552785396576,   0.10936068743467331,   0.036000773310661316,
                                        0.0075987582094967365, 0.001028380123898387};

R FN(orc_fused_loss)(const R *pred, const R *gt, int H, int W, R ssim_weight, R *image_grad, int threads) {
  const R C1 = (R)(0.01f * 1.0f) * (R)(0.01f * 1.0f), C2 = (R)(0.03f * 1.0f) * (R)(0.03f * 1.0f);
  const size_t npx = (size_t)H * W;
  R *hx = (R *)malloc(npx * 3 * 5 * sizeof(R));       /* horizontal pass: sumX, sumX2, sumY, sumY2, sumXY */
  R *dmu = (R *)malloc(npx * 3 * sizeof(R)), *ds1 = (R *)malloc(npx * 3 * sizeof(R)), *ds12 = (R *)malloc(npx * 3 * sizeof(R));
  R *hb = (R *)malloc(npx * 3 * 3 * sizeof(R));
  R w11[11];
  for (int k = 0; k < 11; ++k) w11[k] = (R)(float)FN(kGauss11)[k];
  (void)threads;
  double total = 0.0;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        R sX = 0, sX2 = 0, sY = 0, sY2 = 0, sXY = 0;
        for (int d = 1; d <= 5; ++d) {
          const R w = w11[5 - d];
          const int xl = x - d < 0 ? 0 : x - d, xr = x + d > W - 1 ? W - 1 : x + d;
          const R Xl = pred[((size_t)y * W + xl) * 3 + c], Yl = gt[((size_t)y * W + xl) * 3 + c];
          const R Xr = pred[((size_t)y * W + xr) * 3 + c], Yr = gt[((size_t)y * W + xr) * 3 + c];
          sX += (Xl + Xr) * w; sX2 += (Xl * Xl + Xr * Xr) * w; sY += (Yl + Yr) * w; sY2 += (Yl * Yl + Yr * Yr) * w;
          sXY += (Xl * Yl + Xr * Yr) * w;
        }
        const R wc = w11[5], Xc = pred[((size_t)y * W + x) * 3 + c], Yc = gt[((size_t)y * W + x) * 3 + c];
        sX += Xc * wc; sX2 += Xc * Xc * wc; sY += Yc * wc; sY2 += Yc * Yc * wc; sXY += Xc * Yc * wc;
        R *o = hx + (((size_t)y * W + x) * 3 + c) * 5;
        o[0] = sX; o[1] = sX2; o[2] = sY; o[3] = sY2; o[4] = sXY;
      }
#pragma omp parallel for schedule(static) reduction(+ : total) num_threads(threads > 0 ? threads : 1)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        R o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
        for (int d = 1; d <= 5; ++d) {
          const R w = w11[5 - d];
          const int yt = y - d < 0 ? 0 : y - d, yb = y + d > H - 1 ? H - 1 : y + d;
          const R *t = hx + (((size_t)yt * W + x) * 3 + c) * 5, *b = hx + (((size_t)yb * W + x) * 3 + c) * 5;
          o0 += (t[0] + b[0]) * w; o1 += (t[1] + b[1]) * w; o2 += (t[2] + b[2]) * w; o3 += (t[3] + b[3]) * w;
          o4 += (t[4] + b[4]) * w;
        }
        const R *ct = hx + (((size_t)y * W + x) * 3 + c) * 5;
        const R wc = w11[5];
        o0 += ct[0] * wc; o1 += ct[1] * wc; o2 += ct[2] * wc; o3 += ct[3] * wc; o4 += ct[4] * wc;
        const R mu1 = o0, mu2 = o2, mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
        const R s1 = o1 - mu1_sq, s2 = o3 - mu2_sq, s12 = o4 - mu1 * mu2;
        const R A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2, Cc = (R)2 * mu1 * mu2 + C1, D = (R)2 * s12 + C2;
        const R ssim = (Cc * D) / (A * B);
        const size_t id = ((size_t)y * W + x) * 3 + c;
        const R l1 = FABS(pred[id] - gt[id]);
        total += (double)(((R)1 - ssim_weight) * l1 + ssim_weight * ((R)1 - ssim));
        const R d_mu1 = ((mu2 * (R)2 * D) / (A * B) - (mu2 * (R)2 * Cc) / (A * B) - (mu1 * (R)2 * Cc * D) / (A * A * B) +
                         (mu1 * (R)2 * Cc * D) / (A * B * B));
        dmu[id] = -ssim_weight * d_mu1;
        ds1[id] = -ssim_weight * ((-Cc * D) / (A * B * B));
        ds12[id] = -ssim_weight * (((R)2 * Cc) / (A * B));
      }
  /* backward: zero-padded separable convolution of the three maps */
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        R a0 = 0, a1 = 0, a2 = 0;
        for (int d = 1; d <= 5; ++d) {
          const R w = w11[5 - d];
          const int xl = x - d, xr = x + d;
          const size_t il = ((size_t)y * W + (xl < 0 ? 0 : xl)) * 3 + c, ir = ((size_t)y * W + (xr > W - 1 ? W - 1 : xr)) * 3 + c;
          const R l0 = xl < 0 ? (R)0 : dmu[il], l1 = xl < 0 ? (R)0 : ds1[il], l2 = xl < 0 ? (R)0 : ds12[il];
          const R r0 = xr >= W ? (R)0 : dmu[ir], r1 = xr >= W ? (R)0 : ds1[ir], r2 = xr >= W ? (R)0 : ds12[ir];
          a0 += (l0 + r0) * w; a1 += (l1 + r1) * w; a2 += (l2 + r2) * w;
        }
        const size_t id = ((size_t)y * W + x) * 3 + c;
        a0 += dmu[id] * w11[5]; a1 += ds1[id] * w11[5]; a2 += ds12[id] * w11[5];
        R *o = hb + id * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
      }
  const R grad_scale = (R)1 / (R)((float)(H * W * 3));
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        R s0 = 0, s1 = 0, s2 = 0;
        for (int d = 1; d <= 5; ++d) {
          const R w = w11[5 - d];
          const int yt = y - d, yb = y + d;
          if (yt >= 0) { const R *t = hb + (((size_t)yt * W + x) * 3 + c) * 3; s0 += t[0] * w; s1 += t[1] * w; s2 += t[2] * w; }
          if (yb < H) { const R *b = hb + (((size_t)yb * W + x) * 3 + c) * 3; s0 += b[0] * w; s1 += b[1] * w; s2 += b[2] * w; }
        }
        const size_t id = ((size_t)y * W + x) * 3 + c;
        const R *ct = hb + id * 3;
        s0 += ct[0] * w11[5]; s1 += ct[1] * w11[5]; s2 += ct[2] * w11[5];
        const R ssim_g = s0 + ((R)2 * pred[id]) * s1 + gt[id] * s2;
        const R l1_g = ((R)1 - ssim_weight) * ((pred[id] > gt[id]) ? (R)1 : (R)-1);
        image_grad[id] = (ssim_g + l1_g) * grad_scale;
      }
  free(hx); free(dmu); free(ds1); free(ds12); free(hb);
  return (R)(total / (double)((size_t)H * W * 3));
}

R FN(orc_psnr)(const R *pred, const R *gt, int H, int W) {  /* cuda/loss.cu:476-525 */
  double acc = 0.0;
  const size_t n = (size_t)H * W * 3;
  for (size_t i = 0; i < n; ++i) { const R d = pred[i] - gt[i]; acc += (double)(d * d); }
  const R mse = (R)(acc / (double)n);
  if (mse == (R)0) return (R)100;
  return (R)10 * (R)log10((double)((R)1 / mse));
}

/* F2  adam_step   cuda/optimizer.cu:6-29   ("next" row f2); NaN gradients count as 0 */
void FN(orc_adam)(R *p, const R *g, R *m, R *v, R lr, R b1, R b2, R eps, R bias1, R bias2, long n) {
  for (long i = 0; i < n; ++i) {
    R gr = g[i];
    if (gr != gr) gr = (R)0;
    const R mi = b1 * m[i] + ((R)1 - b1) * gr;
    const R vi = b2 * v[i] + ((R)1 - b2) * gr * gr;
    const R m_hat = mi / bias1, v_hat = vi / bias2;
    p[i] += -lr * m_hat / (SQRT(v_hat) + eps);
    m[i] = mi; v[i] = vi;
  }
}

#undef SH_C0
#undef SH_C1
#undef SH_C2
#undef SH_C3
#undef SH_C4
#undef SH_C5
#undef SH_C6
#undef SH_C7
#undef SH_C8
#undef SH_C9
#undef FN
#undef FN1
#undef FN2
