// gs_math.h -- per-gaussian device math shared by the stand-alone operators and the fused
// preprocess kernels.  Written for gfx950; compiled with -ffp-contract=off so that the
// operation order below IS the arithmetic (the CPU oracle is built the same way), and the
// render kernels ask for FMAs explicitly where they want them.
//
// Each function states the reference operator it re-implements (semantics only; the code
// here is organised around one-thread-per-gaussian register math with explicit structs,
// not around the reference's kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace gs {

struct Mat34 { float m[12]; };  // rows 0..2 of a row-major 4x4 (view)
struct Mat44 { float m[16]; };

__device__ __forceinline__ Mat34 load_view(const float *__restrict__ view) {
  Mat34 v;
#pragma unroll
  for (int k = 0; k < 12; ++k) v.m[k] = view[k];  // uniform address -> scalar loads
  return v;
}
__device__ __forceinline__ Mat44 load_proj(const float *__restrict__ proj) {
  Mat44 p;
#pragma unroll
  for (int k = 0; k < 16; ++k) p.m[k] = proj[k];
  return p;
}

// float -> int with device semantics made explicit (NaN -> 0, saturating)
__device__ __forceinline__ int f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483647.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

// ---- P1: world -> camera (reference: compute_camera_space_points, cuda/projection.cu:6-45)
__device__ __forceinline__ void camera_space(const Mat34 &v, float wx, float wy, float wz, float &x, float &y,
                                             float &z) {
  x = v.m[0] * wx + v.m[1] * wy + v.m[2] * wz + v.m[3];
  y = v.m[4] * wx + v.m[5] * wy + v.m[6] * wz + v.m[7];
  z = v.m[8] * wx + v.m[9] * wy + v.m[10] * wz + v.m[11];
}

// ---- P2: camera -> pixel (reference: project_to_screen, cuda/projection.cu:47-98)
__device__ __forceinline__ void to_screen(const Mat44 &p, float x, float y, float z, int width, int height, float &u,
                                          float &v) {
  const float x_clip = p.m[0] * x + p.m[1] * y + p.m[2] * z + p.m[3];
  const float y_clip = p.m[4] * x + p.m[5] * y + p.m[6] * z + p.m[7];
  const float w_clip = p.m[12] * x + p.m[13] * y + p.m[14] * z + p.m[15];
  const float x_ndc = x_clip / (w_clip + 1e-6f);
  const float y_ndc = y_clip / (w_clip + 1e-6f);
  u = (x_ndc * 0.5f + 0.5f) * (float)width;
  v = (y_ndc * 0.5f + 0.5f) * (float)height;
}

// ---- K1: keep-mask (reference: cull_gaussians, cuda/culling.cu:70-95)
__device__ __forceinline__ bool keep(float u, float v, float z, float near_thresh, int padding, int width,
                                     int height) {
  return (z >= near_thresh) && (u >= (float)(-1 * padding)) && (u <= (float)(width + padding)) &&
         (v >= (float)(-1 * padding)) && (v <= (float)(height + padding));
}

// ---- G1: covariance from quaternion (w,x,y,z) + log-scale (reference: compute_sigma, cuda/gaussian.cu:6-75)
struct RotScale {
  float R[9];
  float s[3];  // exp(scale)
  float inv_norm;
  float q[4];  // normalised (w,x,y,z)
};

__device__ __forceinline__ RotScale rot_scale(float qw, float qx, float qy, float qz, float sx, float sy, float sz) {
  RotScale o;
  const float norm = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
  o.inv_norm = 1.0f / (norm + 1e-6f);
  const float w = qw * o.inv_norm, x = qx * o.inv_norm, y = qy * o.inv_norm, z = qz * o.inv_norm;
  o.q[0] = w; o.q[1] = x; o.q[2] = y; o.q[3] = z;
  const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y,
              wz = w * z;
  o.R[0] = 1.0f - 2.0f * (y2 + z2); o.R[1] = 2.0f * (xy - wz); o.R[2] = 2.0f * (xz + wy);
  o.R[3] = 2.0f * (xy + wz); o.R[4] = 1.0f - 2.0f * (x2 + z2); o.R[5] = 2.0f * (yz - wx);
  o.R[6] = 2.0f * (xz - wy); o.R[7] = 2.0f * (yz + wx); o.R[8] = 1.0f - 2.0f * (x2 + y2);
  o.s[0] = expf(sx); o.s[1] = expf(sy); o.s[2] = expf(sz);
  return o;
}

__device__ __forceinline__ void sigma_from(const RotScale &rs, float *s /*6*/) {
  const float rs00 = rs.R[0] * rs.s[0], rs10 = rs.R[3] * rs.s[0], rs20 = rs.R[6] * rs.s[0];
  const float rs01 = rs.R[1] * rs.s[1], rs11 = rs.R[4] * rs.s[1], rs21 = rs.R[7] * rs.s[1];
  const float rs02 = rs.R[2] * rs.s[2], rs12 = rs.R[5] * rs.s[2], rs22 = rs.R[8] * rs.s[2];
  s[0] = rs00 * rs00 + rs01 * rs01 + rs02 * rs02;
  s[1] = rs00 * rs10 + rs01 * rs11 + rs02 * rs12;
  s[2] = rs00 * rs20 + rs01 * rs21 + rs02 * rs22;
  s[3] = rs10 * rs10 + rs11 * rs11 + rs12 * rs12;
  s[4] = rs10 * rs20 + rs11 * rs21 + rs12 * rs22;
  s[5] = rs20 * rs20 + rs21 * rs21 + rs22 * rs22;
}

// ---- G2a: perspective Jacobian (reference: compute_projection_jacobian_kernel, cuda/gaussian.cu:177-218)
__device__ __forceinline__ void jacobian(float x, float y, float z, float fx, float fy, float tan_fovx,
                                         float tan_fovy, float *J /*6*/) {
  if (fabsf(z) < 1e-6f) {
    J[0] = J[1] = J[2] = J[3] = J[4] = J[5] = 0.0f;
    return;
  }
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float txtz = x / z, tytz = y / z;
  x = fminf(limx, fmaxf(-limx, txtz)) * z;
  y = fminf(limy, fmaxf(-limy, tytz)) * z;
  J[0] = fx / z; J[1] = 0.0f; J[2] = -(fx * x) / (z * z);
  J[3] = 0.0f; J[4] = fy / z; J[5] = -(fy * y) / (z * z);
}

// M = J W (2x3), V = Sigma M^T (3x2): shared by the conic forward and backward
struct MV { float m[6]; float v[6]; };
__device__ __forceinline__ MV mv_from(const float *J, const float *s, const Mat34 &vw) {
  MV o;
  const float w00 = vw.m[0], w01 = vw.m[1], w02 = vw.m[2], w10 = vw.m[4], w11 = vw.m[5], w12 = vw.m[6],
              w20 = vw.m[8], w21 = vw.m[9], w22 = vw.m[10];
  o.m[0] = J[0] * w00 + J[1] * w10 + J[2] * w20;
  o.m[1] = J[0] * w01 + J[1] * w11 + J[2] * w21;
  o.m[2] = J[0] * w02 + J[1] * w12 + J[2] * w22;
  o.m[3] = J[3] * w00 + J[4] * w10 + J[5] * w20;
  o.m[4] = J[3] * w01 + J[4] * w11 + J[5] * w21;
  o.m[5] = J[3] * w02 + J[4] * w12 + J[5] * w22;
  o.v[0] = s[0] * o.m[0] + s[1] * o.m[1] + s[2] * o.m[2];  // v00
  o.v[1] = s[0] * o.m[3] + s[1] * o.m[4] + s[2] * o.m[5];  // v01
  o.v[2] = s[1] * o.m[0] + s[3] * o.m[1] + s[4] * o.m[2];  // v10
  o.v[3] = s[1] * o.m[3] + s[3] * o.m[4] + s[4] * o.m[5];  // v11
  o.v[4] = s[2] * o.m[0] + s[4] * o.m[1] + s[5] * o.m[2];  // v20
  o.v[5] = s[2] * o.m[3] + s[4] * o.m[4] + s[5] * o.m[5];  // v21
  return o;
}

// ---- G2b: 2-D covariance -> conic + OBB radii (reference: compute_conic_kernel, cuda/gaussian.cu:77-175)
__device__ __forceinline__ void conic_radius(const float *J, const float *s, const Mat34 &vw, float mh_dist,
                                             float *conic /*3*/, float *radius /*4*/) {
  const MV a = mv_from(J, s, vw);
  const float cov00 = a.m[0] * a.v[0] + a.m[1] * a.v[2] + a.m[2] * a.v[4] + 0.3f;
  const float cov01 = a.m[0] * a.v[1] + a.m[1] * a.v[3] + a.m[2] * a.v[5];
  const float cov11 = a.m[3] * a.v[1] + a.m[4] * a.v[3] + a.m[5] * a.v[5] + 0.3f;
  const float det = cov00 * cov11 - cov01 * cov01;
  const float inv_det = 1.0f / det;
  conic[0] = cov11 * inv_det;
  conic[1] = -cov01 * inv_det;
  conic[2] = cov00 * inv_det;
  const float mid = 0.5f * (cov00 + cov11);
  const float lambda_term = sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda1 = mid + lambda_term, lambda2 = mid - lambda_term;
  radius[0] = ceilf(mh_dist * sqrtf(lambda1));
  radius[1] = ceilf(mh_dist * sqrtf(lambda2));  // NaN for lambda2 < 0, kept on purpose
  const float th = 0.5f * atan2f(2.0f * cov01, cov00 - cov11);
  float sn, cs;
  sincosf(th, &sn, &cs);
  radius[2] = sn;
  radius[3] = cs;
}

// ---- S1: real spherical harmonics, l <= 3, index l*l+l+m, no Condon-Shortley phase.
// Replaces sphericart::SphericalHarmonics<float> (cuda/spherical_harmonics.cu:72,89).
// Homogeneous-polynomial form (equals Y_lm on the unit sphere).
#define GS_SH_C0 0.28209479177387814f
#define GS_SH_C1 0.4886025119029199f
#define GS_SH_C2 1.0925484305920792f
#define GS_SH_C3 0.31539156525252005f
#define GS_SH_C4 0.5462742152960396f
#define GS_SH_C5 0.5900435899266435f
#define GS_SH_C6 2.890611442640554f
#define GS_SH_C7 0.4570457994644658f
#define GS_SH_C8 0.3731763325901154f
#define GS_SH_C9 1.445305721320277f

template <int L>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float *Y) {
  Y[0] = GS_SH_C0;
  if constexpr (L >= 1) {
    Y[1] = GS_SH_C1 * y; Y[2] = GS_SH_C1 * z; Y[3] = GS_SH_C1 * x;
  }
  if constexpr (L >= 2) {
    const float xx = x * x, yy = y * y, zz = z * z;
    Y[4] = GS_SH_C2 * (x * y);
    Y[5] = GS_SH_C2 * (y * z);
    Y[6] = GS_SH_C3 * (2.0f * zz - xx - yy);
    Y[7] = GS_SH_C2 * (x * z);
    Y[8] = GS_SH_C4 * (xx - yy);
    if constexpr (L >= 3) {
      Y[9] = GS_SH_C5 * (y * (3.0f * xx - yy));
      Y[10] = GS_SH_C6 * (x * y * z);
      Y[11] = GS_SH_C7 * (y * (4.0f * zz - xx - yy));
      Y[12] = GS_SH_C8 * (z * (2.0f * zz - 3.0f * xx - 3.0f * yy));
      Y[13] = GS_SH_C7 * (x * (4.0f * zz - xx - yy));
      Y[14] = GS_SH_C9 * (z * (xx - yy));
      Y[15] = GS_SH_C5 * (x * (xx - 3.0f * yy));
    }
  }
}

// Cartesian gradient of the same polynomials: d[k] = (dY_k/dx, dY_k/dy, dY_k/dz)
template <int L>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float (*d)[3]) {
  d[0][0] = d[0][1] = d[0][2] = 0.0f;
  if constexpr (L >= 1) {
    d[1][0] = 0; d[1][1] = GS_SH_C1; d[1][2] = 0;
    d[2][0] = 0; d[2][1] = 0; d[2][2] = GS_SH_C1;
    d[3][0] = GS_SH_C1; d[3][1] = 0; d[3][2] = 0;
  }
  if constexpr (L >= 2) {
    const float xx = x * x, yy = y * y, zz = z * z;
    d[4][0] = GS_SH_C2 * y; d[4][1] = GS_SH_C2 * x; d[4][2] = 0;
    d[5][0] = 0; d[5][1] = GS_SH_C2 * z; d[5][2] = GS_SH_C2 * y;
    d[6][0] = GS_SH_C3 * (-2.0f * x); d[6][1] = GS_SH_C3 * (-2.0f * y); d[6][2] = GS_SH_C3 * (4.0f * z);
    d[7][0] = GS_SH_C2 * z; d[7][1] = 0; d[7][2] = GS_SH_C2 * x;
    d[8][0] = GS_SH_C4 * (2.0f * x); d[8][1] = GS_SH_C4 * (-2.0f * y); d[8][2] = 0;
    if constexpr (L >= 3) {
      d[9][0] = GS_SH_C5 * (6.0f * x * y); d[9][1] = GS_SH_C5 * (3.0f * xx - 3.0f * yy); d[9][2] = 0;
      d[10][0] = GS_SH_C6 * (y * z); d[10][1] = GS_SH_C6 * (x * z); d[10][2] = GS_SH_C6 * (x * y);
      d[11][0] = GS_SH_C7 * (-2.0f * x * y); d[11][1] = GS_SH_C7 * (4.0f * zz - xx - 3.0f * yy);
      d[11][2] = GS_SH_C7 * (8.0f * y * z);
      d[12][0] = GS_SH_C8 * (-6.0f * x * z); d[12][1] = GS_SH_C8 * (-6.0f * y * z);
      d[12][2] = GS_SH_C8 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
      d[13][0] = GS_SH_C7 * (4.0f * zz - 3.0f * xx - yy); d[13][1] = GS_SH_C7 * (-2.0f * x * y);
      d[13][2] = GS_SH_C7 * (8.0f * x * z);
      d[14][0] = GS_SH_C9 * (2.0f * x * z); d[14][1] = GS_SH_C9 * (-2.0f * y * z); d[14][2] = GS_SH_C9 * (xx - yy);
      d[15][0] = GS_SH_C5 * (3.0f * xx - 3.0f * yy); d[15][1] = GS_SH_C5 * (-6.0f * x * y); d[15][2] = 0;
    }
  }
}

// view direction (reference: compute_dir_kernel, cuda/spherical_harmonics.cu:8-26)
__device__ __forceinline__ void view_dir(float px, float py, float pz, float cx, float cy, float cz, float &dx,
                                         float &dy, float &dz, float &len) {
  const float fx = px - cx, fy = py - cy, fz = pz - cz;
  len = sqrtf(fx * fx + fy * fy + fz * fz) + 1e-9f;
  dx = fx / len; dy = fy / len; dz = fz / len;
}

// SH -> rgb (reference: compute_rgb_from_sh_kernel, cuda/spherical_harmonics.cu:28-60).
// sh points at this gaussian's (n-1)*3 "rest" coefficients, band0 at its 3 DC coefficients.
template <int L>
__device__ __forceinline__ void sh_to_rgb(const float *__restrict__ sh, const float *__restrict__ band0, float dx,
                                          float dy, float dz, float *rgb) {
  constexpr int n = (L + 1) * (L + 1);
  float Y[n];
  sh_basis<L>(dx, dy, dz, Y);
  float r = band0[0] * Y[0] + 0.5f, g = band0[1] * Y[0] + 0.5f, b = band0[2] * Y[0] + 0.5f;
#pragma unroll
  for (int k = 0; k < n - 1; ++k) {
    r += sh[3 * k + 0] * Y[k + 1];
    g += sh[3 * k + 1] * Y[k + 1];
    b += sh[3 * k + 2] * Y[k + 1];
  }
  rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

// ---- B1 pieces: coarse rectangle + OBB/tile separating-axis test
// (reference: coarse_binning_kernel cuda/culling.cu:209-224, compute_obb :148-165, split_axis_test :97-146)
struct TileRect { int x0, x1, y0, y1; };

__device__ __forceinline__ TileRect coarse_rect(float u, float v, float r_major, int ntx, int nty) {
  const int radius_tiles = f2i_sat(ceilf(r_major * 0.0625f)) + 1;
  const int ptx = f2i_sat(floorf(u / 16.0f));
  const int pty = f2i_sat(floorf(v / 16.0f));
  long long sx = (long long)ptx - radius_tiles, ex = (long long)ptx + radius_tiles + 1;
  long long sy = (long long)pty - radius_tiles, ey = (long long)pty + radius_tiles + 1;
  if (sx < 0) sx = 0;
  if (ex > ntx) ex = ntx;
  if (sy < 0) sy = 0;
  if (ey > nty) ey = nty;
  if (ex < sx) ex = sx;
  if (ey < sy) ey = sy;
  TileRect r;
  r.x0 = (int)sx; r.x1 = (int)ex; r.y0 = (int)sy; r.y1 = (int)ey;
  return r;
}

struct Obb {
  float c[8];            // 4 corners (x,y)
  float mnx, mxx, mny, mxy;
  float a2x, a2y, mn2, mx2;  // major axis + projected OBB interval
  float a3x, a3y, mn3, mx3;  // minor axis + projected OBB interval
};

__device__ __forceinline__ Obb make_obb(float u, float v, float r_major, float r_minor, float sin_t, float cos_t) {
  Obb o;
  const float v1x = r_major * cos_t, v1y = r_major * sin_t, v2x = -r_minor * sin_t, v2y = r_minor * cos_t;
  o.c[0] = u - v1x - v2x; o.c[1] = v - v1y - v2y;
  o.c[2] = u + v1x - v2x; o.c[3] = v + v1y - v2y;
  o.c[4] = u - v1x + v2x; o.c[5] = v - v1y + v2y;
  o.c[6] = u + v1x + v2x; o.c[7] = v + v1y + v2y;
  o.mnx = fminf(fminf(o.c[0], o.c[2]), fminf(o.c[4], o.c[6]));
  o.mxx = fmaxf(fmaxf(o.c[0], o.c[2]), fmaxf(o.c[4], o.c[6]));
  o.mny = fminf(fminf(o.c[1], o.c[3]), fminf(o.c[5], o.c[7]));
  o.mxy = fmaxf(fmaxf(o.c[1], o.c[3]), fmaxf(o.c[5], o.c[7]));
  o.a2x = o.c[2] - o.c[0]; o.a2y = o.c[3] - o.c[1];
  {
    const float pr = o.a2x * o.c[2] + o.a2y * o.c[3], pl = o.a2x * o.c[0] + o.a2y * o.c[1];
    o.mn2 = fminf(pr, pl); o.mx2 = fmaxf(pr, pl);
  }
  o.a3x = o.c[2] - o.c[6]; o.a3y = o.c[3] - o.c[7];
  {
    const float pt = o.a3x * o.c[2] + o.a3y * o.c[3], pb = o.a3x * o.c[6] + o.a3y * o.c[7];
    o.mn3 = fminf(pt, pb); o.mx3 = fmaxf(pt, pb);
  }
  return o;
}

// tile AABB = [16tx,16(tx+1)] x [16ty,16(ty+1)], closed.  NaN corners fail every comparison -> "intersects".
__device__ __forceinline__ bool obb_hits_tile(const Obb &o, int tx, int ty) {
  const float l = (float)tx * 16.0f, r = (float)(tx + 1) * 16.0f, t = (float)ty * 16.0f, b = (float)(ty + 1) * 16.0f;
  if (o.mnx > r || o.mxx < l) return false;
  if (o.mny > b || o.mxy < t) return false;
  {
    const float tl = o.a2x * l + o.a2y * t, tr = o.a2x * r + o.a2y * t, bl = o.a2x * l + o.a2y * b,
                br = o.a2x * r + o.a2y * b;
    const float mn = fminf(fminf(tl, tr), fminf(bl, br)), mx = fmaxf(fmaxf(tl, tr), fmaxf(bl, br));
    if (mn > o.mx2 || mx < o.mn2) return false;
  }
  {
    const float tl = o.a3x * l + o.a3y * t, tr = o.a3x * r + o.a3y * t, bl = o.a3x * l + o.a3y * b,
                br = o.a3x * r + o.a3y * b;
    const float mn = fminf(fminf(tl, tr), fminf(bl, br)), mx = fmaxf(fmaxf(tl, tr), fmaxf(bl, br));
    if (mn > o.mx3 || mx < o.mn3) return false;
  }
  return true;
}

// The part of the coarse rectangle whose tiles can pass the two axis-aligned tests of obb_hits_tile: tile tx passes
// them iff 16(tx+1) >= mnx and 16 tx <= mxx, i.e. ceil(mnx/16) - 1 <= tx <= floor(mxx/16) (the scalings are exact).
// Tiles outside fail for certain, so looping over this span with the full test changes no membership; NaN bounds keep
// the whole rectangle (fmaxf / fminf drop the NaN), matching the "NaN passes" rule above.  Typical gaussians need 4-6
// candidates instead of the reference rectangle's 25.
__device__ __forceinline__ TileRect obb_span(const Obb &o, const TileRect &r) {
  TileRect s;
  s.x0 = (int)fminf(fmaxf((float)r.x0, ceilf(o.mnx * 0.0625f) - 1.0f), (float)r.x1);
  s.x1 = (int)fmaxf(fminf((float)(r.x1 - 1), floorf(o.mxx * 0.0625f)), (float)(r.x0 - 1)) + 1;
  s.y0 = (int)fminf(fmaxf((float)r.y0, ceilf(o.mny * 0.0625f) - 1.0f), (float)r.y1);
  s.y1 = (int)fmaxf(fminf((float)(r.y1 - 1), floorf(o.mxy * 0.0625f)), (float)(r.y0 - 1)) + 1;
  return s;
}

// A value of one lane, for a lane index that is uniform across the wave: v_readlane instead of a ds_bpermute round
// trip through the LDS.
__device__ __forceinline__ float lane_value(float x, int uniform_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), uniform_lane));
}
__device__ __forceinline__ int lane_value(int x, int uniform_lane) { return __builtin_amdgcn_readlane(x, uniform_lane); }

// 64 lanes walking the tiles p = lane, lane + 64, ... of a span `sh` tiles high in column-major order (tile p is column
// p / sh, row p % sh): the quotients come from two float divisions per walk and a carry per step instead of an integer
// division per tile.  (lane / sh and 64 / sh are exact through IEEE float division: for integers a <= 64, sh <= 2^14
// the quotient is either an integer or at least 1/sh away from one.)
struct SpanWalk {
  int col, row, dcol, drow, sh;
};
__device__ __forceinline__ SpanWalk span_walk(int lane, int sh) {
  SpanWalk w;
  w.sh = sh;
  w.col = (int)((float)lane / (float)sh);
  w.row = lane - w.col * sh;
  w.dcol = (int)(64.0f / (float)sh);
  w.drow = 64 - w.dcol * sh;
  return w;
}
__device__ __forceinline__ void span_step(SpanWalk &w) {
  w.col += w.dcol;
  w.row += w.drow;
  if (w.row >= w.sh) { w.row -= w.sh; ++w.col; }
}

// monotone map float -> uint32 (ascending float order == ascending unsigned order)
__device__ __forceinline__ unsigned int float_sort_bits(float z) {
  const unsigned int b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ================================ backward pieces ======================================

// ---- Q1 (reference: project_to_screen_backward_kernel, cuda/projection_backward.cu:6-75): returns the increment
__device__ __forceinline__ void to_screen_bwd(const Mat44 &p, float x, float y, float z, float gu, float gv, int width,
                                              int height, float &dx, float &dy, float &dz) {
  const float x_clip = p.m[0] * x + p.m[1] * y + p.m[2] * z + p.m[3];
  const float y_clip = p.m[4] * x + p.m[5] * y + p.m[6] * z + p.m[7];
  const float w_clip = p.m[12] * x + p.m[13] * y + p.m[14] * z + p.m[15];
  dx = dy = dz = 0.0f;
  if (fabsf(w_clip) < 1e-6f) return;
  const float w_inv = 1.0f / w_clip, w_inv2 = w_inv * w_inv;
  const float dx_ndc = gu * (float)width * 0.5f, dy_ndc = gv * (float)height * 0.5f;
  const float dx_clip = dx_ndc * w_inv, dy_clip = dy_ndc * w_inv;
  const float dw_clip = -dx_ndc * x_clip * w_inv2 - dy_ndc * y_clip * w_inv2;
  const float dz_clip = 0.0f;
  dx = p.m[0] * dx_clip + p.m[4] * dy_clip + p.m[8] * dz_clip + p.m[12] * dw_clip;
  dy = p.m[1] * dx_clip + p.m[5] * dy_clip + p.m[9] * dz_clip + p.m[13] * dw_clip;
  dz = p.m[2] * dx_clip + p.m[6] * dy_clip + p.m[10] * dz_clip + p.m[14] * dw_clip;
}

// ---- Q2 (reference: compute_camera_space_points_backward_kernel, cuda/projection_backward.cu:95-137)
__device__ __forceinline__ void camera_space_bwd(const Mat34 &v, float gx, float gy, float gz, float &dx, float &dy,
                                                 float &dz) {
  dx = v.m[0] * gx + v.m[4] * gy + v.m[8] * gz;
  dy = v.m[1] * gx + v.m[5] * gy + v.m[9] * gz;
  dz = v.m[2] * gx + v.m[6] * gy + v.m[10] * gz;
}

// ---- H1 (reference: compute_projection_jacobian_backward_kernel, cuda/gaussian_backward.cu:6-78)
__device__ __forceinline__ void jacobian_bwd(float x, float y, float z, float fx, float fy, float tan_fovx,
                                             float tan_fovy, const float *dJ, float &dx, float &dy, float &dz) {
  dx = dy = dz = 0.0f;
  if (fabsf(z) < 1e-6f) return;
  const float z_inv = 1.0f / (z + 1e-6f), z_inv2 = z_inv * z_inv, z_inv3 = z_inv2 * z_inv;
  const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
  const float txtz = x * z_inv, tytz = y * z_inv;
  const float dJ00 = dJ[0], dJ02 = dJ[2], dJ11 = dJ[4], dJ12 = dJ[5];
  dz += dJ00 * (-fx * z_inv2);
  if (fabsf(txtz) <= limx) {
    dx += dJ02 * (-fx * z_inv2);
    dz += dJ02 * (2.0f * fx * x * z_inv3);
  } else {
    const float cx = (txtz > 0.0f ? limx : -limx);
    dz += dJ02 * (fx * cx * z_inv2);
  }
  dz += dJ11 * (-fy * z_inv2);
  if (fabsf(tytz) <= limy) {
    dy += dJ12 * (-fy * z_inv2);
    dz += dJ12 * (2.0f * fy * y * z_inv3);
  } else {
    const float cy = (tytz > 0.0f ? limy : -limy);
    dz += dJ12 * (fy * cy * z_inv2);
  }
}

// ---- H2 (reference: conic_backward_kernel, cuda/gaussian_backward.cu:97-248): increments for J_grad[6], sigma_grad[6]
__device__ __forceinline__ void conic_bwd(const float *J, const float *s, const Mat34 &vw, const float *c,
                                          const float *dc, float *dJ /*6*/, float *dS /*6*/) {
  const MV a = mv_from(J, s, vw);
  const float *m = a.m, *v = a.v;
  const float t00 = c[0] * dc[0] + c[1] * dc[1], t01 = c[0] * dc[1] + c[1] * dc[2];
  const float t10 = c[1] * dc[0] + c[2] * dc[1], t11 = c[1] * dc[1] + c[2] * dc[2];
  const float d00 = -(t00 * c[0] + t01 * c[1]);
  const float d01 = -(t00 * c[1] + t01 * c[2]);
  const float d11 = -(t10 * c[1] + t11 * c[2]);
  const float dv00 = d00 * m[0] + d01 * m[3], dv01 = d01 * m[0] + d11 * m[3];
  const float dv10 = d00 * m[1] + d01 * m[4], dv11 = d01 * m[1] + d11 * m[4];
  const float dv20 = d00 * m[2] + d01 * m[5], dv21 = d01 * m[2] + d11 * m[5];
  dS[0] = dv00 * m[0] + dv01 * m[3];
  dS[1] = dv00 * m[1] + dv01 * m[4] + dv10 * m[0] + dv11 * m[3];
  dS[2] = dv00 * m[2] + dv01 * m[5] + dv20 * m[0] + dv21 * m[3];
  dS[3] = dv10 * m[1] + dv11 * m[4];
  dS[4] = dv10 * m[2] + dv11 * m[5] + dv20 * m[1] + dv21 * m[4];
  dS[5] = dv20 * m[2] + dv21 * m[5];
  const float dmc00 = d00 * v[0] + d01 * v[1], dmc01 = d00 * v[2] + d01 * v[3], dmc02 = d00 * v[4] + d01 * v[5];
  const float dmc10 = d01 * v[0] + d11 * v[1], dmc11 = d01 * v[2] + d11 * v[3], dmc12 = d01 * v[4] + d11 * v[5];
  const float dmv00 = dv00 * s[0] + dv10 * s[1] + dv20 * s[2], dmv01 = dv00 * s[1] + dv10 * s[3] + dv20 * s[4],
              dmv02 = dv00 * s[2] + dv10 * s[4] + dv20 * s[5];
  const float dmv10 = dv01 * s[0] + dv11 * s[1] + dv21 * s[2], dmv11 = dv01 * s[1] + dv11 * s[3] + dv21 * s[4],
              dmv12 = dv01 * s[2] + dv11 * s[4] + dv21 * s[5];
  const float dm00 = dmc00 + dmv00, dm01 = dmc01 + dmv01, dm02 = dmc02 + dmv02;
  const float dm10 = dmc10 + dmv10, dm11 = dmc11 + dmv11, dm12 = dmc12 + dmv12;
  const float w00 = vw.m[0], w01 = vw.m[1], w02 = vw.m[2], w10 = vw.m[4], w11 = vw.m[5], w12 = vw.m[6],
              w20 = vw.m[8], w21 = vw.m[9], w22 = vw.m[10];
  dJ[0] = dm00 * w00 + dm01 * w01 + dm02 * w02;
  dJ[1] = dm00 * w10 + dm01 * w11 + dm02 * w12;
  dJ[2] = dm00 * w20 + dm01 * w21 + dm02 * w22;
  dJ[3] = dm10 * w00 + dm11 * w01 + dm12 * w02;
  dJ[4] = dm10 * w10 + dm11 * w11 + dm12 * w12;
  dJ[5] = dm10 * w20 + dm11 * w21 + dm12 * w22;
}

// ---- H3 (reference: sigma_backward_kernel, cuda/gaussian_backward.cu:271-415): dQ[4], dS[3] (overwrite semantics)
__device__ __forceinline__ void sigma_bwd(const RotScale &rs, const float *g /*6*/, float *dQ, float *dSc) {
  const float *Rm = rs.R;
  const float Sx = rs.s[0], Sy = rs.s[1], Sz = rs.s[2];
  const float w = rs.q[0], x = rs.q[1], y = rs.q[2], z = rs.q[3];
  float M[9];
  M[0] = Rm[0] * Sx; M[1] = Rm[1] * Sy; M[2] = Rm[2] * Sz;
  M[3] = Rm[3] * Sx; M[4] = Rm[4] * Sy; M[5] = Rm[5] * Sz;
  M[6] = Rm[6] * Sx; M[7] = Rm[7] * Sy; M[8] = Rm[8] * Sz;
  float dSg[9];
  dSg[0] = g[0]; dSg[1] = 0.5f * g[1]; dSg[2] = 0.5f * g[2];
  dSg[3] = 0.5f * g[1]; dSg[4] = g[3]; dSg[5] = 0.5f * g[4];
  dSg[6] = 0.5f * g[2]; dSg[7] = 0.5f * g[4]; dSg[8] = g[5];
  float dM[9], dR[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dM[3 * r + c] = 2.0f * (dSg[3 * r] * M[c] + dSg[3 * r + 1] * M[3 + c] + dSg[3 * r + 2] * M[6 + c]);
  dR[0] = dM[0] * Sx; dR[1] = dM[1] * Sy; dR[2] = dM[2] * Sz;
  dR[3] = dM[3] * Sx; dR[4] = dM[4] * Sy; dR[5] = dM[5] * Sz;
  dR[6] = dM[6] * Sx; dR[7] = dM[7] * Sy; dR[8] = dM[8] * Sz;
  const float dsx = Rm[0] * dM[0] + Rm[3] * dM[3] + Rm[6] * dM[6];
  const float dsy = Rm[1] * dM[1] + Rm[4] * dM[4] + Rm[7] * dM[7];
  const float dsz = Rm[2] * dM[2] + Rm[5] * dM[5] + Rm[8] * dM[8];
  dSc[0] = dsx * Sx; dSc[1] = dsy * Sy; dSc[2] = dsz * Sz;
  float dw = 0.0f, dx = 0.0f, dy = 0.0f, dz = 0.0f;
  dw += dR[1] * (-2.0f * z) + dR[2] * (2.0f * y);
  dx += dR[1] * (2.0f * y) + dR[2] * (2.0f * z);
  dy += dR[0] * (-4.0f * y) + dR[1] * (2.0f * x) + dR[2] * (2.0f * w);
  dz += dR[0] * (-4.0f * z) + dR[1] * (-2.0f * w) + dR[2] * (2.0f * x);
  dw += dR[3] * (2.0f * z) + dR[5] * (-2.0f * x);
  dx += dR[3] * (2.0f * y) + dR[4] * (-4.0f * x) + dR[5] * (-2.0f * w);
  dy += dR[3] * (2.0f * x) + dR[5] * (2.0f * z);
  dz += dR[3] * (2.0f * w) + dR[4] * (-4.0f * z) + dR[5] * (2.0f * y);
  dw += dR[6] * (-2.0f * y) + dR[7] * (2.0f * x);
  dx += dR[6] * (2.0f * z) + dR[7] * (2.0f * w) + dR[8] * (-4.0f * x);
  dy += dR[6] * (-2.0f * w) + dR[7] * (2.0f * z) + dR[8] * (-4.0f * y);
  dz += dR[6] * (2.0f * x) + dR[7] * (2.0f * y);
  const float dot = w * dw + x * dx + y * dy + z * dz;
  dQ[0] = rs.inv_norm * (dw - dot * w);
  dQ[1] = rs.inv_norm * (dx - dot * x);
  dQ[2] = rs.inv_norm * (dy - dot * y);
  dQ[3] = rs.inv_norm * (dz - dot * z);
}

// ---- S2 (reference: compute_sh_gradients_kernel, cuda/spherical_harmonics_backward.cu:28-166)
// Writes sh_grad (n-1)*3 and band0_grad 3 (overwrite), returns the xyz increment.
// kStoreGrad = false (r06, the backward that applies Adam itself): the coefficient gradients are not stored -- their
// consumer rebuilds gr[c] * Y[k + 1] where it needs them -- and `sh` stays as it was.
template <int L, bool kStoreGrad = true>
// sh_grad may be the same row as sh (coefficient k is read before its gradient is written).
__device__ __forceinline__ void sh_bwd(const float *sh, const float *__restrict__ band0, float px,
                                       float py, float pz, float cx, float cy, float cz, const float *gr,
                                       float *sh_grad, float *__restrict__ band0_grad, float &ox,
                                       float &oy, float &oz) {
  constexpr int n = (L + 1) * (L + 1);
  float ux, uy, uz, len;
  view_dir(px, py, pz, cx, cy, cz, ux, uy, uz, len);
  float Y[n];
  float dY[n][3];
  sh_basis<L>(ux, uy, uz, Y);
  sh_basis_grad<L>(ux, uy, uz, dY);
  band0_grad[0] = gr[0] * Y[0]; band0_grad[1] = gr[1] * Y[0]; band0_grad[2] = gr[2] * Y[0];
  float dRx = 0, dGx = 0, dBx = 0, dRy = 0, dGy = 0, dBy = 0, dRz = 0, dGz = 0, dBz = 0;
  {
    const float R0 = band0[0], G0 = band0[1], B0 = band0[2];
    dRx += dY[0][0] * R0; dGx += dY[0][0] * G0; dBx += dY[0][0] * B0;
    dRy += dY[0][1] * R0; dGy += dY[0][1] * G0; dBy += dY[0][1] * B0;
    dRz += dY[0][2] * R0; dGz += dY[0][2] * G0; dBz += dY[0][2] * B0;
  }
#pragma unroll
  for (int k = 0; k < n - 1; ++k) {
    const float yv = Y[k + 1];
    const float Ri = sh[3 * k], Gi = sh[3 * k + 1], Bi = sh[3 * k + 2];
    if constexpr (kStoreGrad) { sh_grad[3 * k] = gr[0] * yv; sh_grad[3 * k + 1] = gr[1] * yv; sh_grad[3 * k + 2] = gr[2] * yv; }
    const float ddx = dY[k + 1][0], ddy = dY[k + 1][1], ddz = dY[k + 1][2];
    dRx += ddx * Ri; dGx += ddx * Gi; dBx += ddx * Bi;
    dRy += ddy * Ri; dGy += ddy * Gi; dBy += ddy * Bi;
    dRz += ddz * Ri; dGz += ddz * Gi; dBz += ddz * Bi;
  }
  const float tx = gr[0] * dRx + gr[1] * dGx + gr[2] * dBx;
  const float ty = gr[0] * dRy + gr[1] * dGy + gr[2] * dBy;
  const float tz = gr[0] * dRz + gr[1] * dGz + gr[2] * dBz;
  const float dot = tx * ux + ty * uy + tz * uz;
  ox = (tx - dot * ux) / len;
  oy = (ty - dot * uy) / len;
  oz = (tz - dot * uz) / len;
}

// ---- f2: one element of the reference's adam_kernel (cuda/optimizer.cu:6-29): NaN gradients count as 0, bias-corrected
// moments, the update in the reference's order of operations.  On values, so that callers decide where they live.
__device__ __forceinline__ void adam_values(float &p, float &m, float &v, float g, float lr, float b1, float b2, float eps,
                                            float bias1, float bias2) {
  if (g != g) g = 0.0f;
  const float mi = b1 * m + (1.0f - b1) * g;
  const float vi = b2 * v + (1.0f - b2) * g * g;
  const float m_hat = mi / bias1, v_hat = vi / bias2;
  p += -lr * m_hat / (sqrtf(v_hat) + eps);
  m = mi;
  v = vi;
}

}  // namespace gs
