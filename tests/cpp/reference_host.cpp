// reference_host.cpp -- one training view the way the REFERENCE host drives the rasterizer, written against the drop-in
// headers only (include/gsplat_cuda/raster.cuh, cuda_data.cuh, cuda_backward.cuh), at any scene size.
//
// Per iteration, as TrainerImpl::train does (/root/reference/cuda/trainer.cu:1294-1360): a fresh ForwardPassData,
// zero_grads() (cuda/trainer.cu:247-261), rasterize_image(...) (cuda/raster.cu:12-136), then the operator chain of
// TrainerImpl::backward_pass (cuda/trainer.cu:941-1012) -- eight compact_masked_array calls and the seven stand-alone
// backward operators on the raw arrays -- with dL/dimage given (the trainer gets it from fused_loss; here it is an input so
// that the gradients can be compared with the oracle's).
//
//   reference_host <scene.bin> <result.bin> <iterations> [keep]
//
// scene.bin / result.bin: 3dgs_amd/scene_io.py.  `keep` reuses one ForwardPassData for all iterations (NOT what the
// reference does; it separates the cost of its per-iteration vector allocations from the rest).  Prints one JSON line
// with the wall-clock milliseconds per iteration and per half.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <thrust/copy.h>
#include <thrust/fill.h>
#include <thrust/host_vector.h>

#include "gsplat_cuda/cuda_backward.cuh"
#include "gsplat_cuda/raster.cuh"

// stand-ins for the reference's Eigen-based host types: rasterize_image reads exactly these members
struct Camera { uint64_t width, height; std::vector<double> params; };
struct Vec3 { double v[3]; double operator[](int i) const { return v[i]; } };
struct Image { Vec3 campos; Vec3 CamPos() const { return campos; } };
struct ConfigParameters { double near_thresh = 0.3, mh_dist = 3.0; int cull_mask_padding = 100; };

namespace {

struct Scene {
  int N = 0, W = 0, H = 0, L = 0;
  float fx = 0, fy = 0, campos[3] = {0, 0, 0}, near_thresh = 0, mh_dist = 0, bg = 0;
  int padding = 0;
  std::vector<float> view, proj, xyz, rgb, sh, opacity, scale, quaternion, grad_image;
};

void read_exact(FILE *f, void *dst, size_t bytes, const char *what) {
  if (bytes && std::fread(dst, 1, bytes, f) != bytes) {
    std::fprintf(stderr, "reference_host: short read (%s)\n", what);
    std::exit(2);
  }
}
void read_floats(FILE *f, std::vector<float> &v, size_t count, const char *what) {
  v.resize(count);
  read_exact(f, v.data(), count * sizeof(float), what);
}

Scene load_scene(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "reference_host: cannot open %s\n", path); std::exit(2); }
  int32_t head[8];
  read_exact(f, head, sizeof(head), "header");
  if (head[0] != 0x31485347) { std::fprintf(stderr, "reference_host: %s is not a scene file\n", path); std::exit(2); }
  Scene s;
  s.N = head[1]; s.W = head[2]; s.H = head[3]; s.L = head[4]; s.padding = head[5];
  float fl[8];
  read_exact(f, fl, sizeof(fl), "scalars");
  s.fx = fl[0]; s.fy = fl[1]; s.campos[0] = fl[2]; s.campos[1] = fl[3]; s.campos[2] = fl[4];
  s.near_thresh = fl[5]; s.mh_dist = fl[6]; s.bg = fl[7];
  const size_t N = (size_t)s.N, rest = (size_t)((s.L + 1) * (s.L + 1) - 1) * 3;
  read_floats(f, s.view, 16, "view"); read_floats(f, s.proj, 16, "proj");
  read_floats(f, s.xyz, N * 3, "xyz"); read_floats(f, s.rgb, N * 3, "rgb"); read_floats(f, s.sh, N * rest, "sh");
  read_floats(f, s.opacity, N, "opacity"); read_floats(f, s.scale, N * 3, "scale");
  read_floats(f, s.quaternion, N * 4, "quaternion");
  read_floats(f, s.grad_image, (size_t)s.W * s.H * 3, "grad_image");
  std::fclose(f);
  return s;
}

template <class V> auto raw(V &v) { return thrust::raw_pointer_cast(v.data()); }
template <class V> auto raw(const V &v) { return thrust::raw_pointer_cast(v.data()); }

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// `profile` runs: every call is followed by a device synchronisation and its wall time goes to a named slot
struct Laps {
  bool on = false;
  double last = 0.0;
  std::vector<std::pair<std::string, double>> slots;
  void start() { if (on) { (void)hipDeviceSynchronize(); last = now_ms(); } }
  void lap(const char *name) {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const double t = now_ms();
    for (auto &s : slots)
      if (s.first == name) { s.second += t - last; last = t; return; }
    slots.emplace_back(name, t - last);
    last = t;
  }
};

// the training state of the reference's TrainerImpl that the two calls touch
struct Host {
  Laps laps;
  CudaDataManager cuda;
  int num_gaussians, l_max;
  Camera camera;
  Image image;
  ConfigParameters config;
  thrust::device_vector<float> d_grad_image;

  Host(const Scene &s)
      : cuda((size_t)s.N), num_gaussians(s.N), l_max(s.L),
        camera{(uint64_t)s.W, (uint64_t)s.H, {s.fx, s.fy, s.W / 2.0, s.H / 2.0}},
        image{{{s.campos[0], s.campos[1], s.campos[2]}}}, d_grad_image(s.grad_image) {
    config.near_thresh = s.near_thresh; config.mh_dist = s.mh_dist; config.cull_mask_padding = s.padding;
    thrust::copy(s.xyz.begin(), s.xyz.end(), cuda.gaussians.d_xyz.begin());
    thrust::copy(s.rgb.begin(), s.rgb.end(), cuda.gaussians.d_rgb.begin());
    thrust::copy(s.sh.begin(), s.sh.end(), cuda.gaussians.d_sh.begin());  // packed at the current band's stride
    thrust::copy(s.opacity.begin(), s.opacity.end(), cuda.gaussians.d_opacity.begin());
    thrust::copy(s.scale.begin(), s.scale.end(), cuda.gaussians.d_scale.begin());
    thrust::copy(s.quaternion.begin(), s.quaternion.end(), cuda.gaussians.d_quaternion.begin());
    thrust::copy(s.view.begin(), s.view.end(), cuda.camera.d_view.begin());
    thrust::copy(s.proj.begin(), s.proj.end(), cuda.camera.d_proj.begin());
  }

  void zero_grads() {  // cuda/trainer.cu:247-261
    const size_t n = (size_t)num_gaussians;
    auto &g = cuda.gradients;
    thrust::fill_n(g.d_grad_xyz.begin(), n * 3, 0.0f); thrust::fill_n(g.d_grad_rgb.begin(), n * 3, 0.0f);
    thrust::fill_n(g.d_grad_sh.begin(), n * 45, 0.0f); thrust::fill_n(g.d_grad_opacity.begin(), n, 0.0f);
    thrust::fill_n(g.d_grad_scale.begin(), n * 3, 0.0f); thrust::fill_n(g.d_grad_quaternion.begin(), n * 4, 0.0f);
    thrust::fill_n(g.d_grad_conic.begin(), n * 3, 0.0f); thrust::fill_n(g.d_grad_uv.begin(), n * 2, 0.0f);
    thrust::fill_n(g.d_grad_J.begin(), n * 6, 0.0f); thrust::fill_n(g.d_grad_sigma.begin(), n * 6, 0.0f);
    thrust::fill_n(g.d_grad_xyz_c.begin(), n * 3, 0.0f); thrust::fill_n(g.d_grad_precompute_rgb.begin(), n * 3, 0.0f);
  }

  void forward(ForwardPassData &pass, float bg) {
    rasterize_image(num_gaussians, camera, image, config, cuda.camera, cuda.gaussians, pass, bg, l_max);
  }

  // cuda/trainer.cu:941-1012 behind fused_loss
  void backward_pass(ForwardPassData &pass, float bg) {
    const int W = (int)camera.width, H = (int)camera.height, M = (int)pass.num_culled;
    auto &prm = cuda.gaussians;
    auto &g = cuda.gradients;
    auto uv_sel = compact_masked_array<2>(pass.d_uv, pass.d_mask, M);
    auto opacity_sel = compact_masked_array<1>(prm.d_opacity, pass.d_mask, M);
    auto xyz_c_sel = compact_masked_array<3>(pass.d_xyz_c, pass.d_mask, M);
    auto quaternion_sel = compact_masked_array<4>(prm.d_quaternion, pass.d_mask, M);
    auto scale_sel = compact_masked_array<3>(prm.d_scale, pass.d_mask, M);
    auto xyz_sel = compact_masked_array<3>(prm.d_xyz, pass.d_mask, M);
    auto rgb_sel = compact_masked_array<3>(prm.d_rgb, pass.d_mask, M);
    laps.lap("compact_masked_array x7 (auto)");
    thrust::device_vector<float> sh_sel;
    if (l_max == 1) sh_sel = compact_masked_array<9>(prm.d_sh, pass.d_mask, M);
    else if (l_max == 2) sh_sel = compact_masked_array<24>(prm.d_sh, pass.d_mask, M);
    else if (l_max == 3) sh_sel = compact_masked_array<45>(prm.d_sh, pass.d_mask, M);
    const float3 campos = make_float3((float)image.campos[0], (float)image.campos[1], (float)image.campos[2]);
    laps.lap("compact_masked_array<45> into a thrust::device_vector (trainer.cu:950-960)");

    render_image_backward(raw(uv_sel), raw(opacity_sel), raw(pass.d_conic), raw(pass.d_precomputed_rgb), bg,
                          raw(pass.d_sorted_gaussians), raw(pass.d_splat_start_end_idx_by_tile_idx),
                          raw(pass.d_splats_per_pixel), raw(pass.d_weight_per_pixel), raw(d_grad_image), W, H,
                          raw(g.d_grad_precompute_rgb), raw(g.d_grad_opacity), raw(g.d_grad_uv), raw(g.d_grad_conic));
    laps.lap("render_image_backward");
    precompute_spherical_harmonics_backward(raw(xyz_sel), raw(rgb_sel), raw(sh_sel), campos, raw(g.d_grad_precompute_rgb),
                                            l_max, M, raw(g.d_grad_sh), raw(g.d_grad_rgb), raw(g.d_grad_xyz));
    laps.lap("precompute_spherical_harmonics_backward");
    compute_conic_backward(raw(pass.d_J), raw(pass.d_sigma), raw(cuda.camera.d_view), raw(pass.d_conic),
                           raw(g.d_grad_conic), M, raw(g.d_grad_J), raw(g.d_grad_sigma));
    laps.lap("compute_conic_backward");
    const float fx = (float)camera.params[0], fy = (float)camera.params[1];
    const float fov_x = 2.0f * std::atan(camera.width / (2.0f * fx)), fov_y = 2.0f * std::atan(camera.height / (2.0f * fy));
    const float tan_fovx = std::tan(fov_x * 0.5f), tan_fovy = std::tan(fov_y * 0.5f);
    compute_projection_jacobian_backward(raw(xyz_c_sel), fx, fy, tan_fovx, tan_fovy, raw(g.d_grad_J), M, raw(g.d_grad_xyz_c));
    laps.lap("compute_projection_jacobian_backward");
    compute_sigma_backward(raw(quaternion_sel), raw(scale_sel), raw(g.d_grad_sigma), M, raw(g.d_grad_quaternion),
                           raw(g.d_grad_scale));
    laps.lap("compute_sigma_backward");
    project_to_screen_backward(raw(xyz_c_sel), raw(cuda.camera.d_proj), raw(g.d_grad_uv), M, W, H, raw(g.d_grad_xyz_c));
    laps.lap("project_to_screen_backward");
    compute_camera_space_points_backward(raw(xyz_sel), raw(cuda.camera.d_view), raw(g.d_grad_xyz_c), M, raw(g.d_grad_xyz));
    laps.lap("compute_camera_space_points_backward");
  }
};

template <class V> void write_floats(FILE *f, const V &dev, size_t count) {
  thrust::host_vector<float> h(dev.begin(), dev.begin() + count);
  if (count && std::fwrite(h.data(), sizeof(float), count, f) != count) { std::fprintf(stderr, "reference_host: write failed\n"); std::exit(2); }
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: reference_host <scene.bin> <result.bin|-> <iterations> [keep]\n"); return 2; }
  const Scene s = load_scene(argv[1]);
  const int iters = std::atoi(argv[3]);
  const bool keep = argc > 4 && std::strcmp(argv[4], "keep") == 0;
  Host host(s);
  ForwardPassData kept;
  size_t M = 0, S = 0;
  auto iteration = [&](bool sync_between, double *fwd_ms, double *bwd_ms) {
    ForwardPassData fresh;  // cuda/trainer.cu:1295: a new ForwardPassData every iteration
    ForwardPassData &pass = keep ? kept : fresh;
    const double t0 = now_ms();
    host.zero_grads();
    host.forward(pass, s.bg);
    if (sync_between) (void)hipDeviceSynchronize();
    const double t1 = now_ms();
    host.backward_pass(pass, s.bg);
    if (sync_between) (void)hipDeviceSynchronize();
    const double t2 = now_ms();
    if (fwd_ms) *fwd_ms += t1 - t0;
    if (bwd_ms) *bwd_ms += t2 - t1;
    M = pass.num_culled; S = pass.d_sorted_gaussians.size();
  };
  for (int k = 0; k < 2; ++k) iteration(false, nullptr, nullptr);  // warm-up: workspace, scratch, clocks
  (void)hipDeviceSynchronize();
  const double t0 = now_ms();
  for (int k = 0; k < iters; ++k) iteration(false, nullptr, nullptr);
  (void)hipDeviceSynchronize();
  const double per_iter = (now_ms() - t0) / (iters > 0 ? iters : 1);
  double fwd_ms = 0.0, bwd_ms = 0.0;
  for (int k = 0; k < iters; ++k) iteration(true, &fwd_ms, &bwd_ms);
  // per-call breakdown: every call followed by a device synchronisation
  host.laps.on = true;
  for (int k = 0; k < iters; ++k) {
    ForwardPassData fresh;
    ForwardPassData &pass = keep ? kept : fresh;
    host.laps.start();
    host.zero_grads();
    host.laps.lap("zero_grads");
    host.forward(pass, s.bg);
    host.laps.lap("rasterize_image");
    host.backward_pass(pass, s.bg);
    host.laps.start();
  }
  host.laps.on = false;
  std::string per_call = "{";
  for (auto &sl : host.laps.slots) {
    char buf[160];
    std::snprintf(buf, sizeof(buf), "%s\"%s\": %.4f", per_call.size() > 1 ? ", " : "", sl.first.c_str(), sl.second / (iters > 0 ? iters : 1));
    per_call += buf;
  }
  per_call += "}";

  if (std::strcmp(argv[2], "-") != 0) {  // one more iteration whose results are kept for the parity check
    ForwardPassData pass;
    host.zero_grads();
    host.forward(pass, s.bg);
    host.backward_pass(pass, s.bg);
    (void)hipDeviceSynchronize();
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) { std::fprintf(stderr, "reference_host: cannot write %s\n", argv[2]); return 2; }
    const size_t Mo = pass.num_culled, rest = (size_t)((s.L + 1) * (s.L + 1) - 1) * 3;
    const int32_t head[8] = {0x31524847, s.N, (int32_t)Mo, s.W, s.H, s.L, (int32_t)pass.d_sorted_gaussians.size(), 0};
    std::fwrite(head, sizeof(head), 1, f);
    write_floats(f, pass.d_image_buffer, (size_t)s.W * s.H * 3);
    auto &g = host.cuda.gradients;
    write_floats(f, g.d_grad_xyz, Mo * 3); write_floats(f, g.d_grad_rgb, Mo * 3); write_floats(f, g.d_grad_sh, Mo * rest);
    write_floats(f, g.d_grad_opacity, Mo); write_floats(f, g.d_grad_scale, Mo * 3); write_floats(f, g.d_grad_quaternion, Mo * 4);
    write_floats(f, g.d_grad_conic, Mo * 3); write_floats(f, g.d_grad_uv, Mo * 2); write_floats(f, g.d_grad_J, Mo * 6);
    write_floats(f, g.d_grad_sigma, Mo * 6); write_floats(f, g.d_grad_xyz_c, Mo * 3); write_floats(f, g.d_grad_precompute_rgb, Mo * 3);
    std::fclose(f);
  }
  std::printf("{\"iterations\": %d, \"forward_pass_data\": \"%s\", \"ms_per_iteration\": %.4f, "
              "\"ms_zero_grads_and_rasterize_image\": %.4f, \"ms_backward_pass\": %.4f, \"ms_per_call_synchronised\": %s, "
              "\"pool_bytes\": %zu, \"num_gaussians\": %d, \"num_culled\": %zu, \"sorted_list_capacity\": %zu}\n",
              iters, keep ? "kept across iterations" : "fresh per iteration (cuda/trainer.cu:1295)", per_iter,
              fwd_ms / (iters > 0 ? iters : 1), bwd_ms / (iters > 0 ? iters : 1), per_call.c_str(), gsplat_pool_bytes(0),
              s.N, M, S);
  return 0;
}
