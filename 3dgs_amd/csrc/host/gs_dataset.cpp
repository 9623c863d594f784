// gs_dataset.cpp -- libgsplat_host.so: COLMAP binary readers, camera geometry, the config subset and the PLY writer
// (include/gsplat_host.h).  Dependency-free restatement of the behaviour of src/colmap.cpp and src/utils.cpp.
#include "../../../include/gsplat_host.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

struct ModelInfo { const char *name; int num_params; };
const ModelInfo kModels[11] = {{"SIMPLE_PINHOLE", 3}, {"PINHOLE", 4}, {"SIMPLE_RADIAL", 4}, {"RADIAL", 5},
                               {"OPENCV", 8}, {"OPENCV_FISHEYE", 8}, {"FULL_OPENCV", 12}, {"FOV", 5},
                               {"SIMPLE_RADIAL_FISHEYE", 4}, {"RADIAL_FISHEYE", 5}, {"THIN_PRISM_FISHEYE", 12}};

// whole file in memory + a bounds-checked cursor: a truncated file is a parse error, never a wild read
struct Reader {
  std::vector<unsigned char> buf;
  size_t pos = 0;
  bool open(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
  }
  template <typename T> bool get(T &v) {
    if (pos + sizeof(T) > buf.size()) return false;
    memcpy(&v, buf.data() + pos, sizeof(T));
    pos += sizeof(T);
    return true;
  }
};

}  // namespace

extern "C" {

const char *gsplat_host_last_error(void) { return g_error; }

const char *gsplat_colmap_model_name(int model_id) {
  return (model_id >= 0 && model_id < 11) ? kModels[model_id].name : nullptr;
}

int gsplat_colmap_read_cameras(const char *path, int downsample_factor, gsplat_colmap_camera *out, size_t capacity,
                               size_t *count) {
  if (!path || !count) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_colmap_read_cameras: null argument");
  if (downsample_factor < 1) return fail(GSPLAT_HOST_ERR_INVALID_ARG, "downsample_factor must be >= 1");
  Reader r;
  if (!r.open(path)) return fail(GSPLAT_HOST_ERR_IO, "could not open %s", path);
  uint64_t n;
  if (!r.get(n)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: missing camera count", path);
  for (uint64_t i = 0; i < n; ++i) {
    gsplat_colmap_camera c;
    memset(&c, 0, sizeof(c));
    uint64_t width, height;  // COLMAP stores 64-bit sizes
    if (!r.get(c.id) || !r.get(c.model_id) || !r.get(width) || !r.get(height))
      return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated camera %llu", path, (unsigned long long)i);
    if (c.model_id != 0 && c.model_id != 1)
      return fail(GSPLAT_HOST_ERR_UNSUPPORTED, "%s: camera %d uses model %d; only SIMPLE_PINHOLE / PINHOLE are supported",
                  path, c.id, c.model_id);
    c.num_params = kModels[c.model_id].num_params;
    for (int k = 0; k < c.num_params; ++k) {
      if (!r.get(c.params[k])) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated camera parameters", path);
      c.params[k] /= (double)downsample_factor;
    }
    c.height = (int)std::round(height / (float)downsample_factor);
    c.width = (int)std::round(width / (float)downsample_factor);
    if (out) {
      if (i >= capacity) return fail(GSPLAT_HOST_ERR_CAPACITY, "%s holds %llu cameras, capacity %zu", path,
                                     (unsigned long long)n, capacity);
      out[i] = c;
    }
  }
  *count = (size_t)n;
  return GSPLAT_HOST_OK;
}

int gsplat_colmap_read_images(const char *path, const char *img_root_dir, int downsample_factor,
                              gsplat_colmap_image *out, size_t capacity, size_t *count, double *xys,
                              int64_t *point3d_ids, size_t points_capacity, size_t *points_count) {
  if (!path || !count || !points_count) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_colmap_read_images: null argument");
  if (downsample_factor < 1) return fail(GSPLAT_HOST_ERR_INVALID_ARG, "downsample_factor must be >= 1");
  Reader r;
  if (!r.open(path)) return fail(GSPLAT_HOST_ERR_IO, "could not open %s", path);
  uint64_t n;
  if (!r.get(n)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: missing image count", path);
  std::string prefix = img_root_dir ? img_root_dir : "";
  prefix += downsample_factor > 1 ? "images_" + std::to_string(downsample_factor) : std::string("images");
  prefix += "/";
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    gsplat_colmap_image im;
    memset(&im, 0, sizeof(im));
    if (!r.get(im.id)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated image %llu", path, (unsigned long long)i);
    for (int k = 0; k < 4; ++k)
      if (!r.get(im.qvec[k])) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated image pose", path);
    for (int k = 0; k < 3; ++k)
      if (!r.get(im.tvec[k])) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated image pose", path);
    if (!r.get(im.camera_id)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated image pose", path);
    std::string name = prefix;
    char ch;
    while (r.get(ch) && ch != '\0') name += ch;
    if (name.size() >= sizeof(im.name)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: image name longer than %zu bytes", path, sizeof(im.name) - 1);
    memcpy(im.name, name.c_str(), name.size() + 1);
    uint64_t np;
    if (!r.get(np)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: missing 2D point count", path);
    im.first_point2d = total;
    im.num_points2d = np;
    for (uint64_t j = 0; j < np; ++j) {
      double x, y;
      int64_t pid;
      if (!r.get(x) || !r.get(y) || !r.get(pid)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated 2D points", path);
      if (xys || point3d_ids) {
        if (total + j >= points_capacity)
          return fail(GSPLAT_HOST_ERR_CAPACITY, "%s: more than %zu 2D points", path, points_capacity);
        if (xys) { xys[2 * (total + j)] = x; xys[2 * (total + j) + 1] = y; }
        if (point3d_ids) point3d_ids[total + j] = pid;
      }
    }
    total += np;
    if (out) {
      if (i >= capacity) return fail(GSPLAT_HOST_ERR_CAPACITY, "%s holds %llu images, capacity %zu", path,
                                     (unsigned long long)n, capacity);
      out[i] = im;
    }
  }
  *count = (size_t)n;
  *points_count = (size_t)total;
  return GSPLAT_HOST_OK;
}

int gsplat_colmap_read_points3d(const char *path, gsplat_colmap_point3d *out, size_t capacity, size_t *count,
                                int *image_ids, int *point2d_idxs, size_t track_capacity, size_t *track_count) {
  if (!path || !count || !track_count) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_colmap_read_points3d: null argument");
  Reader r;
  if (!r.open(path)) return fail(GSPLAT_HOST_ERR_IO, "could not open %s", path);
  uint64_t n;
  if (!r.get(n)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: missing point count", path);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    gsplat_colmap_point3d p;
    memset(&p, 0, sizeof(p));
    if (!r.get(p.id) || !r.get(p.xyz[0]) || !r.get(p.xyz[1]) || !r.get(p.xyz[2]) || !r.get(p.rgb[0]) ||
        !r.get(p.rgb[1]) || !r.get(p.rgb[2]) || !r.get(p.error) || !r.get(p.track_length))
      return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated point %llu", path, (unsigned long long)i);
    p.first_track = total;
    for (uint64_t j = 0; j < p.track_length; ++j) {
      int img, idx;
      if (!r.get(img) || !r.get(idx)) return fail(GSPLAT_HOST_ERR_PARSE, "%s: truncated track", path);
      if (image_ids || point2d_idxs) {
        if (total + j >= track_capacity)
          return fail(GSPLAT_HOST_ERR_CAPACITY, "%s: more than %zu track elements", path, track_capacity);
        if (image_ids) image_ids[total + j] = img;
        if (point2d_idxs) point2d_idxs[total + j] = idx;
      }
    }
    total += p.track_length;
    if (out) {
      if (i >= capacity) return fail(GSPLAT_HOST_ERR_CAPACITY, "%s holds %llu points, capacity %zu", path,
                                     (unsigned long long)n, capacity);
      out[i] = p;
    }
  }
  *count = (size_t)n;
  *track_count = (size_t)total;
  return GSPLAT_HOST_OK;
}

void gsplat_qvec_to_rotmat(const double q[4], double R[9]) {
  // unit-quaternion formula (Eigen::Quaterniond::toRotationMatrix does not normalise either)
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

void gsplat_camera_position(const double q[4], const double t[3], double out[3]) {
  double R[9];
  gsplat_qvec_to_rotmat(q, R);
  for (int k = 0; k < 3; ++k) out[k] = -(R[k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]);
}

int gsplat_scene_extent(const double *qvecs, const double *tvecs, size_t n, double *out) {
  if (!out || (n && (!qvecs || !tvecs))) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_scene_extent: null argument");
  *out = 0.0;
  if (n == 0) return GSPLAT_HOST_OK;
  std::vector<double> c(3 * n);
  double mean[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    gsplat_camera_position(qvecs + 4 * i, tvecs + 3 * i, &c[3 * i]);
    for (int k = 0; k < 3; ++k) mean[k] += c[3 * i + k];
  }
  for (int k = 0; k < 3; ++k) mean[k] /= (double)n;
  double best = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double dx = c[3 * i] - mean[0], dy = c[3 * i + 1] - mean[1], dz = c[3 * i + 2] - mean[2];
    best = std::max(best, std::sqrt(dx * dx + dy * dy + dz * dz));
  }
  *out = best;
  return GSPLAT_HOST_OK;
}

// ---------------------------------------------------------------------------------------------- config
static std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

int gsplat_parse_config(const char *path, gsplat_config *out) {
  if (!path || !out) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_parse_config: null argument");
  std::ifstream f(path);
  if (!f) return fail(GSPLAT_HOST_ERR_IO, "could not open %s", path);
  std::map<std::string, std::string> kv;
  std::string line;
  while (std::getline(f, line)) {
    bool in_quote = false;
    char quote = 0;
    size_t cut = line.size();
    for (size_t i = 0; i < line.size(); ++i) {  // strip comments outside quotes
      const char ch = line[i];
      if (in_quote) { if (ch == quote) in_quote = false; }
      else if (ch == '"' || ch == '\'') { in_quote = true; quote = ch; }
      else if (ch == '#') { cut = i; break; }
    }
    line = trim(line.substr(0, cut));
    if (line.empty() || line == "---") continue;
    const size_t colon = line.find(':');
    if (colon == std::string::npos) return fail(GSPLAT_HOST_ERR_PARSE, "%s: not a 'key: value' line: %s", path, line.c_str());
    std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
    if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front())
      val = val.substr(1, val.size() - 2);
    kv[key] = val;
  }
  memset(out, 0, sizeof(*out));
  int rc = GSPLAT_HOST_OK;
  auto need = [&](const char *k) -> const std::string * {
    auto it = kv.find(k);
    if (it == kv.end()) {
      if (!rc) rc = fail(GSPLAT_HOST_ERR_PARSE, "Missing required parameter in YAML file: %s", k);
      return nullptr;
    }
    return &it->second;
  };
  auto S = [&](const char *k, char *dst, size_t cap) {
    if (const std::string *v = need(k)) {
      if (v->size() >= cap) { if (!rc) rc = fail(GSPLAT_HOST_ERR_PARSE, "%s: value of %s is too long", path, k); return; }
      memcpy(dst, v->c_str(), v->size() + 1);
    }
  };
  auto I = [&](const char *k, int &dst) {
    if (const std::string *v = need(k)) {
      char *end = nullptr;
      const long x = strtol(v->c_str(), &end, 10);
      if (end == v->c_str() || *end) { if (!rc) rc = fail(GSPLAT_HOST_ERR_PARSE, "%s: %s is not an integer: %s", path, k, v->c_str()); return; }
      dst = (int)x;
    }
  };
  auto D = [&](const char *k, double &dst) {
    if (const std::string *v = need(k)) {
      char *end = nullptr;
      const double x = strtod(v->c_str(), &end);
      if (end == v->c_str() || *end) { if (!rc) rc = fail(GSPLAT_HOST_ERR_PARSE, "%s: %s is not a number: %s", path, k, v->c_str()); return; }
      dst = x;
    }
  };
  auto B = [&](const char *k, int &dst) {
    if (const std::string *v = need(k)) {
      if (*v == "true" || *v == "True" || *v == "TRUE" || *v == "yes" || *v == "on") dst = 1;
      else if (*v == "false" || *v == "False" || *v == "FALSE" || *v == "no" || *v == "off") dst = 0;
      else if (!rc) rc = fail(GSPLAT_HOST_ERR_PARSE, "%s: %s is not a boolean: %s", path, k, v->c_str());
    }
  };
  gsplat_config &c = *out;
  S("dataset_path", c.dataset_path, sizeof(c.dataset_path)); S("output_dir", c.output_dir, sizeof(c.output_dir));
  I("downsample_factor", c.downsample_factor); I("print_interval", c.print_interval); I("num_iters", c.num_iters);
  D("ssim_frac", c.ssim_frac); I("test_eval_interval", c.test_eval_interval); I("test_split_ratio", c.test_split_ratio);
  D("initial_opacity", c.initial_opacity); I("initial_scale_num_neighbors", c.initial_scale_num_neighbors);
  D("initial_scale_factor", c.initial_scale_factor); D("max_initial_scale", c.max_initial_scale);
  D("near_thresh", c.near_thresh); D("mh_dist", c.mh_dist); I("cull_mask_padding", c.cull_mask_padding);
  D("base_lr", c.base_lr); D("xyz_lr_multiplier_init", c.xyz_lr_multiplier_init);
  D("xyz_lr_multiplier_final", c.xyz_lr_multiplier_final); D("quat_lr_multiplier", c.quat_lr_multiplier);
  D("scale_lr_multiplier", c.scale_lr_multiplier); D("opacity_lr_multiplier", c.opacity_lr_multiplier);
  D("rgb_lr_multiplier", c.rgb_lr_multiplier); D("sh_lr_multiplier", c.sh_lr_multiplier);
  B("use_background", c.use_background); I("use_background_end", c.use_background_end);
  I("reset_opacity_interval", c.reset_opacity_interval); D("reset_opacity_value", c.reset_opacity_value);
  I("reset_opacity_start", c.reset_opacity_start); I("reset_opacity_end", c.reset_opacity_end);
  B("use_sh_precompute", c.use_sh_precompute); I("max_sh_band", c.max_sh_band);
  I("add_sh_band_interval", c.add_sh_band_interval);
  B("use_split", c.use_split); B("use_clone", c.use_clone); B("use_delete", c.use_delete);
  I("adaptive_control_start", c.adaptive_control_start); I("adaptive_control_end", c.adaptive_control_end);
  I("adaptive_control_interval", c.adaptive_control_interval); I("max_gaussians", c.max_gaussians);
  D("delete_opacity_threshold", c.delete_opacity_threshold); D("uv_grad_threshold", c.uv_grad_threshold);
  D("split_scale_factor", c.split_scale_factor);
  return rc;
}

// ---------------------------------------------------------------------------------------------- PLY
int gsplat_save_ply(const char *path, size_t n, int sh_floats, const float *xyz, const float *rgb, const float *sh,
                    const float *opacity, const float *scale, const float *quaternion) {
  if (!path) return fail(GSPLAT_HOST_ERR_NULL, "gsplat_save_ply: null path");
  if (n && (!xyz || !rgb || !opacity || !scale || !quaternion || (sh_floats > 0 && !sh)))
    return fail(GSPLAT_HOST_ERR_NULL, "gsplat_save_ply: null attribute array");
  if (sh_floats < 0) return fail(GSPLAT_HOST_ERR_INVALID_ARG, "gsplat_save_ply: negative sh_floats");
  FILE *f = fopen(path, "wb");
  if (!f) return fail(GSPLAT_HOST_ERR_IO, "could not open %s for writing", path);
  std::string h = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(n) + "\n";
  for (const char *p : {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"}) h += std::string("property float ") + p + "\n";
  for (int i = 0; i < sh_floats; ++i) h += "property float f_rest_" + std::to_string(i) + "\n";
  for (const char *p : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"})
    h += std::string("property float ") + p + "\n";
  h += "end_header\n";
  bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
  const size_t row = 17 + (size_t)sh_floats;
  std::vector<float> buf(row * 4096);
  for (size_t base = 0; base < n && ok; base += 4096) {
    const size_t m = std::min<size_t>(4096, n - base);
    for (size_t r = 0; r < m; ++r) {
      const size_t i = base + r;
      float *o = &buf[r * row];
      o[0] = xyz[3 * i]; o[1] = xyz[3 * i + 1]; o[2] = xyz[3 * i + 2];
      o[3] = o[4] = o[5] = 0.0f;
      o[6] = rgb[3 * i]; o[7] = rgb[3 * i + 1]; o[8] = rgb[3 * i + 2];
      for (int k = 0; k < sh_floats; ++k) o[9 + k] = sh[(size_t)sh_floats * i + k];
      float *t = o + 9 + sh_floats;
      t[0] = opacity[i];
      t[1] = scale[3 * i]; t[2] = scale[3 * i + 1]; t[3] = scale[3 * i + 2];
      t[4] = quaternion[4 * i + 1]; t[5] = quaternion[4 * i + 2]; t[6] = quaternion[4 * i + 3]; t[7] = quaternion[4 * i];
    }
    ok = fwrite(buf.data(), sizeof(float), m * row, f) == m * row;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? GSPLAT_HOST_OK : fail(GSPLAT_HOST_ERR_IO, "short write to %s", path);
}

}  // extern "C"
