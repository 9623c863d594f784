// cuda_data.cuh -- source-compatible shim for the reference's device-buffer structs and compaction templates
// (reference: include/gsplat_cuda/cuda_data.cuh:11-167, constructors cuda/data.cu:9-107).  The structs keep the
// reference's member names and capacities (they ARE the interface its trainer and tests use); rocThrust provides
// thrust::device_vector.  compact_masked_array / scatter_masked_array forward to libgsplat_hip.so instead of the
// reference's thrust::copy_if / scatter compositions.
#pragma once

#include <thrust/device_vector.h>

#include <cstdio>
#include <cstdlib>
#include <exception>
#include <new>
#include <type_traits>
#include <utility>

#include "hip_compat.h"

#include <thrust/device_ptr.h>
#include <thrust/fill.h>
#include <thrust/host_vector.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/transform_iterator.h>

#ifndef GSPLAT_SHIM_NO_THRUST_FILL
// The reference host clears its gradient vectors with twelve thrust::fill_n calls per iteration (zero_grads,
// cuda/trainer.cu:247-261).  Each is a kernel followed by a stream synchronisation inside thrust (0.27 ms per
// iteration at 1e6 gaussians, a third of it waiting).  This overload -- more specialised than thrust's own template, so
// overload resolution picks it for `thrust::fill_n(vec.begin(), n, 0.0f)` on a thrust::device_vector<float> -- queues
// the fill on the default stream and returns (gsplat_fill_f32: hipMemsetAsync for zero).  Everything the reference does
// afterwards runs on that stream or blocks on it, so the order of effects is unchanged.  -DGSPLAT_SHIM_NO_THRUST_FILL
// switches it off.
namespace thrust {
template <typename Size>
inline detail::normal_iterator<device_ptr<float>> fill_n(detail::normal_iterator<device_ptr<float>> first, Size n,
                                                         const float &value) {
  if (n > 0) gsplat_shim::require_ok(gsplat_fill_f32(raw_pointer_cast(first.base()), (size_t)n, value, 0), "thrust::fill_n");
  return first + n;
}
}  // namespace thrust
#endif

namespace gsplat_shim {
// A device vector for the objects the reference host creates and destroys EVERY iteration: the members of
// ForwardPassData (cuda/trainer.cu:1295 constructs a fresh one per iteration, :1027 copies it by value) and the results
// of compact_masked_array (8 per backward_pass, 16+ per optimizer_step).  As thrust::device_vectors each of them is a
// hipMalloc + a hipFree that synchronises the device: 4 of the 6.5 ms such an iteration took at 1e6 gaussians.  This
// class has the part of device_vector's interface those call sites use -- size / empty / resize / clear / data /
// begin / end / operator[], copy and move, conversion to thrust::device_vector and thrust::host_vector -- on storage
// from the library's block pool (gsplat_pool_alloc: a freed block is handed to the next request of its size class, in
// stream order), so the steady state of a training loop allocates nothing.  Unlike device_vector it does NOT
// value-initialise new elements (every user overwrites them).
// the workspace of the last rasterize_image call (raster.cuh sets it)
inline gsplat_context *&last_context() { static gsplat_context *c = nullptr; return c; }

// The compaction the last forward already computed, if `mask` IS the mask array that forward produced (ForwardPassData's
// d_mask owns that very block since r05): slot of every kept row and the row of every slot, valid until the next
// rasterize_image.  The reference host compacts ~25 arrays per iteration by pass_data.d_mask (cuda/trainer.cu:941-964,
// 1028-1044); with the slots known each is ONE launch and no scan of the mask.
struct known_compaction { const int *slots = nullptr, *rows = nullptr; };
inline known_compaction compaction_of(const unsigned char *mask, size_t n, int num_culled) {
  known_compaction k;
  if (gsplat_context *ctx = last_context()) {
    const unsigned char *m = nullptr;
    const int *slots = nullptr, *c2g = nullptr;
    int N = 0, M = 0;
    if (gsplat_context_last_compaction(ctx, &m, &slots, &c2g, &N, &M) == GSPLAT_OK && m != nullptr && m == mask &&
        (size_t)N == n && M == num_culled) {
      k.slots = slots;
      k.rows = c2g;
    }
  }
  return k;
}

// gathers element e of a compaction on the fly: row e / STRIDE of the compacted array is row rows[e / STRIDE] of the source
template <typename T, int STRIDE> struct gather_rows_fn {
  const T *src;
  const int *rows;
  int stride_rt;
  __host__ __device__ T operator()(unsigned int e) const {
    const unsigned int st = STRIDE > 0 ? (unsigned int)STRIDE : (unsigned int)stride_rt;
    const unsigned int r = e / st;
    return src[(size_t)rows[r] * st + (e - r * st)];
  }
};

// A thrust::device_vector<T> whose n elements are allocated but NOT value-initialised: thrust's own `device_vector v(n)`
// runs a fill kernel over the new elements and synchronises the stream behind it.  The conversions below fill such a
// vector with their own (asynchronous) kernel or copy and move it into the thrust::device_vector<T> the host named --
// the only cost left of `thrust::device_vector<float> d_sh_selected = compact_masked_array<45>(...)`
// (cuda/trainer.cu:950-960) is then the vector's hipMalloc and, when the host destroys it, its hipFree.
// Uses vector_base's protected members m_storage / m_size (a derived class may).  Whether THIS rocThrust has them under
// those names is probed at compile time (raw_thrust_vector_ok: the expressions below in an unevaluated SFINAE context
// of the derived class); when it does not -- a release that renames them -- the conversions fall back to the
// gather-iterator construction by themselves instead of breaking the maintainer's build (r06).
// -DGSPLAT_SHIM_NO_RAW_THRUST_VECTOR forces the fallback.
template <typename T> struct uninitialized_device_vector : thrust::device_vector<T> {
  template <typename U = uninitialized_device_vector,
            typename = decltype(std::declval<U &>().m_storage.allocate(size_t(1))),
            typename = decltype(std::declval<U &>().m_size = size_t(1))>
  static std::true_type probe(int);
  template <typename U = uninitialized_device_vector> static std::false_type probe(...);

  explicit uninitialized_device_vector(size_t n) { reserve_raw<uninitialized_device_vector>(n); }

 private:
  template <typename U> void reserve_raw(size_t n) {  // (a template: only instantiated where the probe succeeded)
    if (n) {
      static_cast<U *>(this)->m_storage.allocate(n);
      static_cast<U *>(this)->m_size = n;
    }
  }
};
#ifdef GSPLAT_SHIM_NO_RAW_THRUST_VECTOR
template <typename T> inline constexpr bool raw_thrust_vector_ok = false;
#else
template <typename T>
inline constexpr bool raw_thrust_vector_ok = decltype(uninitialized_device_vector<T>::template probe<>(0))::value;
#endif

template <typename T> class device_array {
 public:
  using value_type = T;
  using size_type = size_t;
  using pointer = thrust::device_ptr<T>;
  using const_pointer = thrust::device_ptr<const T>;
  using iterator = pointer;
  using const_iterator = const_pointer;

  device_array() = default;
  explicit device_array(size_t n) { resize(n); }
  device_array(const device_array &o) { o.materialize(); assign_raw(o.ptr_, o.size_); }
  device_array(device_array &&o) noexcept : ptr_(o.ptr_), size_(o.size_), cap_(o.cap_), pend_(o.pend_) {
    o.ptr_ = nullptr; o.size_ = o.cap_ = 0; o.pend_.on = false;
  }
  template <typename A> device_array(const thrust::device_vector<T, A> &v) { assign_raw(thrust::raw_pointer_cast(v.data()), v.size()); }
  device_array &operator=(device_array o) noexcept { swap(o); return *this; }
  ~device_array() { release(); }

  void swap(device_array &o) noexcept {
    std::swap(ptr_, o.ptr_); std::swap(size_, o.size_); std::swap(cap_, o.cap_); std::swap(pend_, o.pend_);
  }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  void clear() { size_ = 0; pend_.on = false; }
  // new elements are uninitialised; the first min(old, new) elements are kept
  void resize(size_t n) {
    materialize();
    if (n > cap_) {
      void *fresh = nullptr;
      const int st = gsplat_pool_alloc(&fresh, n * sizeof(T));
      if (st != GSPLAT_OK) throw std::bad_alloc();
      if (size_) (void)hipMemcpyAsync(fresh, ptr_, size_ * sizeof(T), hipMemcpyDeviceToDevice, 0);
      release();
      ptr_ = static_cast<T *>(fresh);
      cap_ = n;
    }
    size_ = n;
  }
  // r05: take over a block of the library's pool that already holds n elements (rasterize_image hands ForwardPassData
  // the forward's own output arrays: gsplat_context_detach_forward_outputs) -- no allocation, no copy
  void adopt(const void *pool_block, size_t n) {
    release();
    ptr_ = static_cast<T *>(const_cast<void *>(pool_block));
    size_ = cap_ = n;
  }
  // r05: a compaction that has not run yet (compact_masked_array of the SH strides: cuda/trainer.cu:950-960 assigns its
  // result to a thrust::device_vector, which the conversion below then fills with ONE gather kernel instead of a
  // compaction into pool storage + a value-initialising allocation + a copy).  Anything else that looks at the elements
  // -- data(), begin(), a copy -- runs the compaction into pool storage first, so the object behaves as if it had run.
  // The source and the mask must stay as they are until then (they do at the reference's call sites).
  void defer_compaction(const T *src, const unsigned char *mask, int N, int stride, int rows, known_compaction known) {
    release();
    size_ = (size_t)rows * (size_t)stride;
    pend_ = pending_t{src, mask, N, stride, rows, true, known};
  }
  pointer data() { materialize(); return pointer(ptr_); }
  const_pointer data() const { materialize(); return const_pointer(ptr_); }
  iterator begin() { materialize(); return pointer(ptr_); }
  iterator end() { materialize(); return pointer(ptr_ + size_); }
  const_iterator begin() const { materialize(); return const_pointer(ptr_); }
  const_iterator end() const { materialize(); return const_pointer(ptr_ + size_); }
  const_iterator cbegin() const { return begin(); }
  const_iterator cend() const { return end(); }
  thrust::device_reference<T> operator[](size_t i) { return *(begin() + i); }
  thrust::device_reference<const T> operator[](size_t i) const { return *(begin() + i); }

  // copies, for the call sites that name thrust's types (`thrust::device_vector<float> d_sh_selected; d_sh_selected =
  // compact_masked_array<45>(...)`, cuda/trainer.cu:950-960; `thrust::host_vector<bool> mask = pass.d_mask`).  The
  // vector is built from an iterator range: thrust allocates and runs one copy kernel, without the value-initialising
  // fill (and its stream synchronisation) that `device_vector v(n)` would add.
  operator thrust::device_vector<T>() const {
    if constexpr (raw_thrust_vector_ok<T>) {
      uninitialized_device_vector<T> v(size_);
      if (size_) {
        T *dst = thrust::raw_pointer_cast(v.data());
        if (pend_.on) {  // the deferred compaction runs straight into the host's vector: no intermediate, no copy
          const pending_t &p = pend_;
          if (p.known.slots)
            require_ok(gsplat_compact_rows_ranked(reinterpret_cast<const float *>(p.src), p.mask, p.known.slots, p.N, p.stride,
                                                  reinterpret_cast<float *>(dst), p.rows, 0), "compact_masked_array");
          else
            require_ok(gsplat_compact_masked_array_bounded(reinterpret_cast<const float *>(p.src), p.mask, p.N, p.stride,
                                                           reinterpret_cast<float *>(dst), p.rows, nullptr, 0),
                       "compact_masked_array");
        } else {
          (void)hipMemcpyAsync(dst, ptr_, size_ * sizeof(T), hipMemcpyDeviceToDevice, 0);
        }
      }
      return thrust::device_vector<T>(std::move(v));
    }
    if (pend_.on && size_ > 0 && size_ < 0xFFFFFFFFull) {  // gather straight into the new vector
      device_array<int> rows;
      const int *rp = pend_.known.rows;  // the forward's compact_to_global, when the mask is the forward's own
      if (!rp) {
        rows.resize((size_t)pend_.rows);
        const int st = gsplat_mask_selected_rows(pend_.mask, pend_.N, thrust::raw_pointer_cast(rows.data()), pend_.rows, 0);
        if (st != GSPLAT_OK) throw std::bad_alloc();
        rp = thrust::raw_pointer_cast(rows.data());
      }
      const thrust::counting_iterator<unsigned int> zero(0u);
      const unsigned int n = (unsigned int)size_;
#define GSPLAT_GATHER_INTO_VECTOR(S)                                                                                   \
  do {                                                                                                                 \
    auto first = thrust::make_transform_iterator(zero, gather_rows_fn<T, S>{pend_.src, rp, pend_.stride});             \
    return thrust::device_vector<T>(first, first + n);                                                                 \
  } while (0)
      switch (pend_.stride) {
        case 9: GSPLAT_GATHER_INTO_VECTOR(9);
        case 24: GSPLAT_GATHER_INTO_VECTOR(24);
        case 45: GSPLAT_GATHER_INTO_VECTOR(45);
        default: GSPLAT_GATHER_INTO_VECTOR(0);
      }
#undef GSPLAT_GATHER_INTO_VECTOR
    }
    return thrust::device_vector<T>(begin(), end());
  }
  operator thrust::host_vector<T>() const {
    materialize();
    thrust::host_vector<T> v(size_);
    if (size_) (void)hipMemcpy(thrust::raw_pointer_cast(v.data()), ptr_, size_ * sizeof(T), hipMemcpyDeviceToHost);
    return v;
  }

 private:
  struct pending_t {
    const T *src = nullptr;
    const unsigned char *mask = nullptr;
    int N = 0, stride = 0, rows = 0;
    bool on = false;
    known_compaction known;
  };
  void materialize() const {
    if (!pend_.on) return;
    const pending_t p = pend_;
    pend_.on = false;
    if (size_ == 0) return;
    void *fresh = nullptr;
    if (gsplat_pool_alloc(&fresh, size_ * sizeof(T)) != GSPLAT_OK) throw std::bad_alloc();
    ptr_ = static_cast<T *>(fresh);
    cap_ = size_;
    static_assert(sizeof(T) == 4 || sizeof(T) == 1 || sizeof(T) == 16, "element sizes of the reference's arrays");
    if (p.known.slots)
      require_ok(gsplat_compact_rows_ranked(reinterpret_cast<const float *>(p.src), p.mask, p.known.slots, p.N, p.stride,
                                            reinterpret_cast<float *>(ptr_), p.rows, 0), "compact_masked_array");
    else
      require_ok(gsplat_compact_masked_array_bounded(reinterpret_cast<const float *>(p.src), p.mask, p.N, p.stride,
                                                     reinterpret_cast<float *>(ptr_), p.rows, nullptr, 0),
                 "compact_masked_array");
  }
  void assign_raw(const T *src, size_t n) {
    resize(n);
    if (n) (void)hipMemcpyAsync(ptr_, src, n * sizeof(T), hipMemcpyDeviceToDevice, 0);
  }
  void release() {
    if (ptr_) (void)gsplat_pool_free(ptr_);
    ptr_ = nullptr;
    size_ = cap_ = 0;
    pend_.on = false;
  }
  mutable T *ptr_ = nullptr;
  size_t size_ = 0;
  mutable size_t cap_ = 0;
  mutable pending_t pend_;
};

template <typename F> inline void alloc_or_exit(const char *what, F &&f) {
  try {
    f();
  } catch (const std::exception &e) {  // cuda/data.cu: print and exit on allocation failure
    std::fprintf(stderr, "CUDA Memory Allocation Error (%s): %s\n", what, e.what());
    std::exit(EXIT_FAILURE);
  }
}
}  // namespace gsplat_shim

struct GaussianParameters {  // SH is allocated at its full capacity (15 coefficients) and packed at the current band
  thrust::device_vector<float> d_xyz, d_rgb, d_sh, d_opacity, d_scale, d_quaternion;
  explicit GaussianParameters(size_t n) {
    gsplat_shim::alloc_or_exit("GaussianParameters", [&] {
      d_xyz.resize(n * 3); d_rgb.resize(n * 3); d_sh.resize(n * 45); d_opacity.resize(n); d_scale.resize(n * 3);
      d_quaternion.resize(n * 4);
    });
  }
};

struct OptimizerParameters {  // Adam moments, zero-initialised
  thrust::device_vector<float> m_grad_xyz, m_grad_rgb, m_grad_sh, m_grad_opacity, m_grad_scale, m_grad_quaternion;
  thrust::device_vector<float> v_grad_xyz, v_grad_rgb, v_grad_sh, v_grad_opacity, v_grad_scale, v_grad_quaternion;
  explicit OptimizerParameters(size_t n) {
    gsplat_shim::alloc_or_exit("OptimizerParameters", [&] {
      m_grad_xyz.assign(n * 3, 0.f); m_grad_rgb.assign(n * 3, 0.f); m_grad_sh.assign(n * 45, 0.f);
      m_grad_opacity.assign(n, 0.f); m_grad_scale.assign(n * 3, 0.f); m_grad_quaternion.assign(n * 4, 0.f);
      v_grad_xyz.assign(n * 3, 0.f); v_grad_rgb.assign(n * 3, 0.f); v_grad_sh.assign(n * 45, 0.f);
      v_grad_opacity.assign(n, 0.f); v_grad_scale.assign(n * 3, 0.f); v_grad_quaternion.assign(n * 4, 0.f);
    });
  }
};

struct GaussianGradients {  // compacted order; the second line are the intermediates of the backward chain
  thrust::device_vector<float> d_grad_xyz, d_grad_rgb, d_grad_sh, d_grad_opacity, d_grad_scale, d_grad_quaternion;
  thrust::device_vector<float> d_grad_conic, d_grad_uv, d_grad_J, d_grad_sigma, d_grad_xyz_c, d_grad_precompute_rgb;
  explicit GaussianGradients(size_t n) {
    gsplat_shim::alloc_or_exit("GaussianGradients", [&] {
      d_grad_xyz.resize(n * 3); d_grad_rgb.resize(n * 3); d_grad_sh.resize(n * 45); d_grad_opacity.resize(n);
      d_grad_scale.resize(n * 3); d_grad_quaternion.resize(n * 4);
      d_grad_conic.resize(n * 3); d_grad_uv.resize(n * 2); d_grad_J.resize(n * 6); d_grad_sigma.resize(n * 6);
      d_grad_xyz_c.resize(n * 3); d_grad_precompute_rgb.resize(n * 3);
    });
  }
};

struct GradientAccumulators {  // density-control statistics, zero-initialised
  thrust::device_vector<float> d_uv_grad_accum;
  thrust::device_vector<int> d_grad_accum_dur;
  explicit GradientAccumulators(size_t n) {
    gsplat_shim::alloc_or_exit("GradientAccumulators", [&] { d_uv_grad_accum.assign(n, 0.f); d_grad_accum_dur.assign(n, 0); });
  }
};

struct CameraParameters {  // row-major 4x4 view and projection matrices
  thrust::device_vector<float> d_view, d_proj;
  CameraParameters() {
    gsplat_shim::alloc_or_exit("CudaDataManager", [&] { d_view.resize(16); d_proj.resize(16); });
  }
};

struct CudaDataManager {  // owns every persistent device buffer of a training run
  const size_t max_gaussians;
  GaussianParameters gaussians;
  OptimizerParameters optimizer;
  GaussianGradients gradients;
  GradientAccumulators accumulators;
  CameraParameters camera;
  explicit CudaDataManager(size_t n) : max_gaussians(n), gaussians(n), optimizer(n), gradients(n), accumulators(n), camera() {}
};

// Per-view outputs of rasterize_image, saved for the backward pass.  Member names, element types and sizes are the
// reference's (cuda_data.cuh:70-86); the containers are gsplat_shim::device_array (above) instead of
// thrust::device_vector, because the reference host constructs this struct afresh every iteration.
struct ForwardPassData {
  size_t num_culled = 0;
  gsplat_shim::device_array<float> d_sigma, d_conic, d_J, d_precomputed_rgb;  // [num_culled, 6 | 3 | 6 | 3]
  gsplat_shim::device_array<float> d_uv, d_xyz_c;                             // [N, 2 | 3], uncompacted
  gsplat_shim::device_array<bool> d_mask;                                     // [N]
  gsplat_shim::device_array<float4> d_radius;                                 // [num_culled]
  gsplat_shim::device_array<int> d_sorted_gaussians, d_splat_start_end_idx_by_tile_idx;
  gsplat_shim::device_array<float> d_image_buffer, d_weight_per_pixel;
  gsplat_shim::device_array<int> d_splats_per_pixel;
};

// compact_masked_array<STRIDE>(source, mask, num_culled): the rows of `source` whose mask entry is set, in order.
// `source` / `mask`: thrust::device_vector or gsplat_shim::device_array.  Like the reference's (cuda_data.cuh:106-127) it
// trusts num_culled (the count rasterize_image reported for this mask) and does not read the count back.
template <int STRIDE, typename Source, typename Mask>
gsplat_shim::device_array<typename Source::value_type> compact_masked_array(const Source &d_source, const Mask &d_mask,
                                                                            int num_culled) {
  using T = typename Source::value_type;
  static_assert(sizeof(T) == 4, "rows are made of 4-byte elements (float / int)");
  static_assert(sizeof(typename Mask::value_type) == 1, "the mask is one byte per row (bool)");
  gsplat_shim::device_array<T> d_selected;
  const unsigned char *mask_ptr = reinterpret_cast<const unsigned char *>(thrust::raw_pointer_cast(d_mask.data()));
  const gsplat_shim::known_compaction known = gsplat_shim::compaction_of(mask_ptr, d_mask.size(), num_culled);
  if constexpr (STRIDE >= 9) {  // the SH strides: deferred (see device_array::defer_compaction)
    d_selected.defer_compaction(thrust::raw_pointer_cast(d_source.data()), mask_ptr, (int)d_mask.size(), STRIDE, num_culled,
                                known);
    return d_selected;
  }
  gsplat_shim::alloc_or_exit("compact_masked_array", [&] { d_selected.resize((size_t)num_culled * STRIDE); });
  const float *src = reinterpret_cast<const float *>(thrust::raw_pointer_cast(d_source.data()));
  float *dst = reinterpret_cast<float *>(thrust::raw_pointer_cast(d_selected.data()));
  if (known.slots)
    gsplat_shim::require_ok(gsplat_compact_rows_ranked(src, mask_ptr, known.slots, (int)d_mask.size(), STRIDE, dst, num_culled, 0),
                            "compact_masked_array");
  else
    gsplat_shim::require_ok(gsplat_compact_masked_array_bounded(src, mask_ptr, (int)d_mask.size(), STRIDE, dst, num_culled,
                                                                nullptr, 0),
                            "compact_masked_array");
  return d_selected;
}

// scatter_masked_array<STRIDE>(compacted, mask, destination): the inverse; rows whose mask entry is clear are untouched.
template <int STRIDE, typename Compacted, typename Mask, typename Destination>
void scatter_masked_array(const Compacted &d_compacted, const Mask &d_mask, Destination &d_destination) {
  using T = typename Destination::value_type;
  static_assert(sizeof(T) == 4 && sizeof(typename Mask::value_type) == 1, "4-byte elements, one-byte mask");
  static_assert(sizeof(typename Compacted::value_type) == 4, "4-byte elements");
  if (d_compacted.size() / STRIDE == 0) return;
  gsplat_shim::require_ok(
      gsplat_scatter_masked_array(reinterpret_cast<const float *>(thrust::raw_pointer_cast(d_compacted.data())),
                                  reinterpret_cast<const unsigned char *>(thrust::raw_pointer_cast(d_mask.data())),
                                  (int)d_mask.size(), STRIDE,
                                  reinterpret_cast<float *>(thrust::raw_pointer_cast(d_destination.data())), 0),
      "scatter_masked_array");
}
