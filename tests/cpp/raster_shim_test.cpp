// Host program written against the reference's raster.cuh / cuda_data.cuh interface: fills a CudaDataManager,
// calls rasterize_image(...) with stand-ins for the reference's Eigen-based Camera / Image / ConfigParameters (the
// shim is a template on those), checks the ForwardPassData it gets back, and exercises the compaction templates
// (tests/cuda_data_test.cpp:38-125).
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

#include <thrust/copy.h>
#include <thrust/fill.h>
#include <thrust/host_vector.h>

#include "gsplat_cuda/raster.cuh"

struct Camera { unsigned long width, height; std::vector<double> params; };
struct Vec3 { double v[3]; double operator[](int i) const { return v[i]; } };
struct Image { Vec3 CamPos() const { return {{0.0, 0.0, 0.0}}; } };
struct ConfigParameters { double near_thresh = 0.3, mh_dist = 3.0; int cull_mask_padding = 100; };

static int failures = 0;
#define EXPECT(cond)                                                       \
  do {                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

// The rocThrust of this image has vector_base::m_storage / m_size: the conversions to thrust::device_vector take the path
// without the value-initialising fill.  (Under -DGSPLAT_SHIM_NO_RAW_THRUST_VECTOR, or with a rocThrust that renames the
// members, the probe is false and the gather-iterator construction is used: tests/test_shim_cpp.py compiles both.)
#ifndef GSPLAT_SHIM_NO_RAW_THRUST_VECTOR
static_assert(gsplat_shim::raw_thrust_vector_ok<float>, "the probe does not find rocThrust's vector_base members");
#else
static_assert(!gsplat_shim::raw_thrust_vector_ok<float>, "the switch must select the fallback");
#endif

int main() {
  const int N = 3, W = 64, H = 48;
  CudaDataManager cuda(8);
  // three gaussians in front of an identity camera, the middle one behind it
  const std::vector<float> xyz = {0.f, 0.f, 5.f, 0.2f, 0.1f, -4.f, -0.5f, 0.3f, 7.f};
  thrust::copy(xyz.begin(), xyz.end(), cuda.gaussians.d_xyz.begin());
  thrust::fill(cuda.gaussians.d_rgb.begin(), cuda.gaussians.d_rgb.end(), 1.0f);
  thrust::fill(cuda.gaussians.d_opacity.begin(), cuda.gaussians.d_opacity.end(), 2.0f);
  thrust::fill(cuda.gaussians.d_scale.begin(), cuda.gaussians.d_scale.end(), std::log(0.2f));
  const std::vector<float> q = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  thrust::copy(q.begin(), q.end(), cuda.gaussians.d_quaternion.begin());
  const float fx = W / (2.0f * std::tan(0.5235988f));
  Camera cam{(unsigned long)W, (unsigned long)H, {fx, fx, W / 2.0, H / 2.0}};
  const float znear = 0.01f, zfar = 100.f, right = (W / (2 * fx)) * znear, top = (H / (2 * fx)) * znear;
  std::vector<float> view = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, proj(16, 0.f);
  proj[0] = znear / right; proj[5] = znear / top; proj[10] = zfar / (zfar - znear); proj[11] = -(zfar * znear) / (zfar - znear);
  proj[14] = 1.f;  // cuda/trainer.cu:1310-1318
  thrust::copy(view.begin(), view.end(), cuda.camera.d_view.begin());
  thrust::copy(proj.begin(), proj.end(), cuda.camera.d_proj.begin());
  ForwardPassData pass;
  rasterize_image(N, cam, Image{}, ConfigParameters{}, cuda.camera, cuda.gaussians, pass, 0.0f, 0);
  (void)hipDeviceSynchronize();
  EXPECT(pass.num_culled == 2);
  thrust::host_vector<bool> mask = pass.d_mask;
  EXPECT(mask[0] && !mask[1] && mask[2]);
  EXPECT(pass.d_sigma.size() == 12 && pass.d_conic.size() == 6 && pass.d_radius.size() == 2);
  EXPECT(pass.d_image_buffer.size() == (size_t)W * H * 3 && pass.d_splats_per_pixel.size() == (size_t)W * H);
  EXPECT(pass.d_splat_start_end_idx_by_tile_idx.size() == 4 * 3 + 1);
  thrust::host_vector<float> img = pass.d_image_buffer;
  const float centre = img[((H / 2) * W + W / 2) * 3];  // gaussian 0 projects onto the image centre, SH band 0 = 1
  EXPECT(centre > 0.5f && std::isfinite(centre));
  EXPECT(img[0] == 0.0f);  // far corner: background 0
  thrust::host_vector<int> ranges = pass.d_splat_start_end_idx_by_tile_idx;
  EXPECT(ranges[0] == 0 && ranges[12] == (int)pass.d_sorted_gaussians.size());
  // compaction templates, tests/cuda_data_test.cpp:47-54, 91-98
  thrust::device_vector<float> src(std::vector<float>{1, 2, 3, 4, 5, 6, 7, 8, 9});
  thrust::device_vector<bool> m(std::vector<bool>{true, false, true});
  thrust::host_vector<float> sel = compact_masked_array<3>(src, m, 2);
  EXPECT(sel.size() == 6 && sel[0] == 1 && sel[3] == 7 && sel[5] == 9);
  thrust::device_vector<float> dst(9, -1.f), comp(std::vector<float>{10, 11, 12, 13, 14, 15});
  scatter_masked_array<3>(comp, m, dst);
  thrust::host_vector<float> hd = dst;
  EXPECT(hd[0] == 10 && hd[2] == 12 && hd[3] == -1 && hd[6] == 13 && hd[8] == 15);
  // gsplat_shim::device_array (what ForwardPassData and compact_masked_array hand out): copy, move, resize keeps the head,
  // conversion to thrust's vectors, storage returned to the pool and taken again
  {
    gsplat_shim::device_array<float> a = compact_masked_array<3>(src, m, 2);
    gsplat_shim::device_array<float> b = a;               // deep copy
    gsplat_shim::device_array<float> c2 = std::move(a);   // a is empty now
    EXPECT(a.empty() && b.size() == 6 && c2.size() == 6);
    b.resize(9);                                          // grows: the first six elements survive
    thrust::host_vector<float> hb = b;
    EXPECT(hb.size() == 9 && hb[0] == 1 && hb[5] == 9);
    thrust::device_vector<float> as_thrust = c2;          // the call sites that name thrust's type (trainer.cu:950-960)
    EXPECT(as_thrust.size() == 6 && (float)as_thrust[3] == 7.f);
    EXPECT((float)c2[4] == 8.f);
    const float *before = thrust::raw_pointer_cast(c2.data());
    c2 = gsplat_shim::device_array<float>();              // back to the pool ...
    gsplat_shim::device_array<float> again(6);            // ... and out again: same size class, same block
    EXPECT(thrust::raw_pointer_cast(again.data()) == before);
    gsplat_shim::device_array<float> from_thrust(as_thrust);
    thrust::host_vector<float> hf = from_thrust;
    EXPECT(hf.size() == 6 && hf[0] == 1 && hf[5] == 9);
  }
  if (failures == 0) std::printf("raster_shim_test: all checks passed\n");
  return failures ? 1 : 0;
}
