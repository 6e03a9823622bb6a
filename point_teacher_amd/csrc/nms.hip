// NMS, rotated IoU / rotated NMS and the step-1 quad rasteriser for gfx950.
// NMS is the classic 64x64 bitmask formulation: one workgroup of 64 threads per
// (row-block, col-block) tile writes a 64-bit suppression word per row; a single
// wavefront then walks the rows in score order keeping the "removed" bit-set in registers
// (one or two 64-bit words per lane) - no host round trip.
#include "pt_common.h"
#include "pt_rotated_iou.h"

namespace pt {

constexpr int NMS_MAXN = 32768;   // nms_pre (2000-3000) x classes (8-9) candidates of the test_cfgs fit

__device__ __forceinline__ float iou_xyxy(const float4 a, const float4 b) {
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  const float ua = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
  return inter / ua;
}

__global__ void box_iou_rotated_kernel(const float* __restrict__ a, const float* __restrict__ b, int M, int N,
                                       int aligned, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = aligned ? M : (long)M * N;
  if (i >= total) return;
  const int m = aligned ? (int)i : (int)(i / N), n = aligned ? (int)i : (int)(i % N);
  float x[5], y[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { x[k] = a[(size_t)m * 5 + k]; y[k] = b[(size_t)n * 5 + k]; }
  out[i] = rotated_iou(x, y);
}

// ------------------------------------------------------------------ bitmask ---
// 256 threads per 64x64 tile: thread (r = tid & 63, q = tid >> 6) tests row r against columns
// [16q, 16q+16); the four partial words of a row are OR-ed through LDS.
template <bool ROT>
__global__ void __launch_bounds__(256)
    nms_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ cls, int N, float thr,
                    unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;  // only the upper triangle is ever read
  const int cols = cdiv(N, 64);
  constexpr int D = ROT ? 5 : 4;
  __shared__ float cbox[64 * 5];
  __shared__ int ccls[64];
  __shared__ unsigned long long part[4][64];
  const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
  if (q == 0) {
    const int j = cb * 64 + r;
    if (j < N) {
#pragma unroll
      for (int k = 0; k < D; ++k) cbox[r * 5 + k] = boxes[(size_t)j * D + k];
      ccls[r] = cls ? cls[j] : 0;
    }
  }
  __syncthreads();
  const int i = rb * 64 + r;
  unsigned long long bits = 0ull;
  if (i < N) {
    float me[5];
#pragma unroll
    for (int k = 0; k < D; ++k) me[k] = boxes[(size_t)i * D + k];
    const int mycls = cls ? cls[i] : 0;
    const int nc = min(64, N - cb * 64);
    const int lo = max(q * 16, (rb == cb) ? r + 1 : 0), hi = min(q * 16 + 16, nc);
    for (int t = lo; t < hi; ++t) {
      if (ccls[t] != mycls) continue;
      float v;
      if (ROT) {
        v = rotated_iou(me, &cbox[t * 5]);
      } else {
        v = iou_xyxy(make_float4(me[0], me[1], me[2], me[3]),
                     make_float4(cbox[t * 5], cbox[t * 5 + 1], cbox[t * 5 + 2], cbox[t * 5 + 3]));
      }
      if (v > thr) bits |= 1ull << t;
    }
  }
  part[q][r] = bits;
  __syncthreads();
  if (q == 0 && i < N) mask[(size_t)i * cols + cb] = part[0][r] | part[1][r] | part[2][r] | part[3][r];
}

// One wavefront: lane l owns the removed-words l, l+64, ..., l+64*(WPL-1) (64*64*WPL candidates).
template <int WPL>
__global__ void __launch_bounds__(64)
    nms_scan_kernel(const unsigned long long* __restrict__ mask, int N, uint8_t* __restrict__ keep) {
  const int cols = cdiv(N, 64);
  const int lane = threadIdx.x;
  unsigned long long r[WPL];
#pragma unroll
  for (int k = 0; k < WPL; ++k) r[k] = 0ull;
  for (int i = 0; i < N; ++i) {
    const int wd = i >> 6;
    unsigned long long src = r[0];
#pragma unroll
    for (int k = 1; k < WPL; ++k) src = ((wd >> 6) == k) ? r[k] : src;      // wd is wave-uniform
    const unsigned lo = __shfl((unsigned)(src & 0xffffffffu), wd & 63, 64);
    const unsigned hi = __shfl((unsigned)(src >> 32), wd & 63, 64);
    const unsigned long long word = ((unsigned long long)hi << 32) | lo;
    const bool removed = (word >> (i & 63)) & 1ull;
    if (lane == 0) keep[i] = removed ? 0 : 1;
    if (!removed) {
      // row i only has valid words for columns >= i/64
#pragma unroll
      for (int k = 0; k < WPL; ++k) {
        const int c = lane + 64 * k;
        if (c < cols && c >= wd) r[k] |= mask[(size_t)i * cols + c];
      }
    }
  }
}

// ------------------------------------------------------------- quad fill -----
__device__ __forceinline__ void fill_one_quad(float* __restrict__ img, int C, int H, int W, const float* __restrict__ quads,
                                              const uint8_t* __restrict__ alive, float value, int q) {
  if (alive && !alive[q]) return;
  long long vx[4], vy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vx[i] = (long long)(int)quads[(size_t)q * 8 + 2 * i];       // astype(np.int32): truncation
    vy[i] = (long long)(int)quads[(size_t)q * 8 + 2 * i + 1];
  }
  long long area2 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) area2 += vx[i] * vy[(i + 1) & 3] - vx[(i + 1) & 3] * vy[i];
  if (area2 == 0) return;
  const long long sgn = area2 > 0 ? 1 : -1;
  long long x0 = vx[0], x1 = vx[0], y0 = vy[0], y1 = vy[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    x0 = vx[i] < x0 ? vx[i] : x0; x1 = vx[i] > x1 ? vx[i] : x1;
    y0 = vy[i] < y0 ? vy[i] : y0; y1 = vy[i] > y1 ? vy[i] : y1;
  }
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > W - 1 ? W - 1 : x1; y1 = y1 > H - 1 ? H - 1 : y1;
  if (x1 < x0 || y1 < y0) return;
  const int bw = (int)(x1 - x0 + 1), bh = (int)(y1 - y0 + 1);
  for (int t = threadIdx.x; t < bw * bh; t += blockDim.x) {
    const long long x = x0 + t % bw, y = y0 + t / bw;
    bool in = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long cr = (vx[(i + 1) & 3] - vx[i]) * (y - vy[i]) - (vy[(i + 1) & 3] - vy[i]) * (x - vx[i]);
      in = in && (cr * sgn >= 0);
    }
    if (in)
      for (int c = 0; c < C; ++c) img[((size_t)c * H + y) * W + x] = value;
  }
}

__global__ void __launch_bounds__(256)
    fill_quads_kernel(float* __restrict__ img, int C, int H, int W, const float* __restrict__ quads,
                      const uint8_t* __restrict__ alive, float value) {
  fill_one_quad(img, C, H, W, quads, alive, value, blockIdx.x);
}

// ---- per-batch forms (r03): the step-1 generator runs one rotated NMS and one rasteriser pass per IMAGE of the batch; each is a
// latency-bound serial scan (150 us for ~600 candidates), so the images' scans run side by side as workgroups of ONE launch.
constexpr int NMS_MAXSEG = 16;
struct NmsSegs {
  int off[NMS_MAXSEG + 1];          // candidate offsets of the segments (images)
  long ws[NMS_MAXSEG];              // word offset of each segment's bit matrix in the workspace
  int n;
};

__global__ void __launch_bounds__(256)
    nms_rotated_mask_seg_kernel(const float* __restrict__ boxes, NmsSegs sg, float thr, unsigned long long* __restrict__ mask) {
  const int s = blockIdx.z, N = sg.off[s + 1] - sg.off[s];
  const int cols = cdiv(N, 64);
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb || cb >= cols) return;
  boxes += (size_t)sg.off[s] * 5;
  mask += sg.ws[s];
  __shared__ float cbox[64 * 5];
  __shared__ unsigned long long part[4][64];
  const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
  if (q == 0) {
    const int j = cb * 64 + r;
    if (j < N) {
#pragma unroll
      for (int k = 0; k < 5; ++k) cbox[r * 5 + k] = boxes[(size_t)j * 5 + k];
    }
  }
  __syncthreads();
  const int i = rb * 64 + r;
  unsigned long long bits = 0ull;
  if (i < N) {
    float me[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) me[k] = boxes[(size_t)i * 5 + k];
    const int nc = min(64, N - cb * 64);
    const int lo = max(q * 16, (rb == cb) ? r + 1 : 0), hi = min(q * 16 + 16, nc);
    for (int t = lo; t < hi; ++t)
      if (rotated_iou(me, &cbox[t * 5]) > thr) bits |= 1ull << t;
  }
  part[q][r] = bits;
  __syncthreads();
  if (q == 0 && i < N) mask[(size_t)i * cols + cb] = part[0][r] | part[1][r] | part[2][r] | part[3][r];
}

__global__ void __launch_bounds__(64)
    nms_scan_seg_kernel(const unsigned long long* __restrict__ mask, NmsSegs sg, uint8_t* __restrict__ keep) {
  const int s = blockIdx.x, N = sg.off[s + 1] - sg.off[s];
  mask += sg.ws[s];
  keep += sg.off[s];
  const int cols = cdiv(N, 64);
  const int lane = threadIdx.x;
  unsigned long long r[2] = {0ull, 0ull};                    // N <= 8192 per segment
  for (int i = 0; i < N; ++i) {
    const int wd = i >> 6;
    const unsigned long long src = (wd >> 6) ? r[1] : r[0];
    const unsigned lo = __shfl((unsigned)(src & 0xffffffffu), wd & 63, 64);
    const unsigned hi = __shfl((unsigned)(src >> 32), wd & 63, 64);
    const bool removed = ((((unsigned long long)hi << 32) | lo) >> (i & 63)) & 1ull;
    if (lane == 0) keep[i] = removed ? 0 : 1;
    if (!removed) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        if (c < cols && c >= wd) r[k] |= mask[(size_t)i * cols + c];
      }
    }
  }
}

// quads of a whole batch: quad q belongs to image img_of[q]; img is [B, C, H, W]
__global__ void __launch_bounds__(256)
    fill_quads_batch_kernel(float* __restrict__ img, int C, int H, int W, const float* __restrict__ quads,
                            const uint8_t* __restrict__ alive, const int32_t* __restrict__ img_of, float value) {
  const int q = blockIdx.x;
  fill_one_quad(img + (size_t)img_of[q] * C * H * W, C, H, W, quads, alive, value, q);
}

}  // namespace pt

using namespace pt;

template <bool ROT>
static int nms_impl(const char* fn, const float* boxes, const int32_t* cls, int N, float thr, uint64_t* ws,
                    uint8_t* keep, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(boxes && ws && keep && N > 0, PT_EINVAL, "%s: bad argument", fn);
  PT_REQUIRE(N <= NMS_MAXN, PT_ELIMIT, "%s: N=%d above %d", fn, N, NMS_MAXN);
  const int cols = cdiv(N, 64);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(nms_mask_kernel<ROT>, dim3(cols, cols), dim3(256), 0, s, boxes, cls, N, thr,
                     reinterpret_cast<unsigned long long*>(ws));
  PT_LAUNCH_CHECK(fn);
  if (N <= 8192)
    hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(ws), N,
                       keep);
  else
    hipLaunchKernelGGL(nms_scan_kernel<8>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(ws), N,
                       keep);
  PT_LAUNCH_CHECK(fn);
  return PT_OK;
}

extern "C" int pt_nms_sorted(const float* boxes, const int32_t* class_id, int N, float iou_thr, uint64_t* mask_ws,
                             uint8_t* keep, void* stream) {
  return nms_impl<false>("pt_nms_sorted", boxes, class_id, N, iou_thr, mask_ws, keep, stream);
}

extern "C" int pt_nms_rotated_sorted(const float* dets, int N, float iou_thr, uint64_t* mask_ws, uint8_t* keep,
                                     void* stream) {
  return nms_impl<true>("pt_nms_rotated_sorted", dets, nullptr, N, iou_thr, mask_ws, keep, stream);
}

extern "C" int pt_box_iou_rotated(const float* a, const float* b, int M, int N, int aligned, float* out,
                                  void* stream) {
  if (M == 0 || N == 0) return PT_OK;
  PT_REQUIRE(a && b && out && M > 0 && N > 0 && (!aligned || M == N), PT_EINVAL, "pt_box_iou_rotated: bad argument");
  const long total = aligned ? M : (long)M * N;
  hipLaunchKernelGGL(box_iou_rotated_kernel, dim3(cdiv(total, 128)), dim3(128), 0, as_stream(stream), a, b, M, N,
                     aligned, out);
  PT_LAUNCH_CHECK("pt_box_iou_rotated");
  return PT_OK;
}

extern "C" int pt_fill_quads(float* img, int C, int H, int W, const float* quads, const uint8_t* alive, int Q,
                             float value, void* stream) {
  if (Q == 0) return PT_OK;
  PT_REQUIRE(img && quads && C > 0 && H > 0 && W > 0 && Q > 0, PT_EINVAL, "pt_fill_quads: bad argument");
  hipLaunchKernelGGL(fill_quads_kernel, dim3(Q), dim3(256), 0, as_stream(stream), img, C, H, W, quads, alive, value);
  PT_LAUNCH_CHECK("pt_fill_quads");
  return PT_OK;
}

extern "C" int pt_nms_rotated_sorted_segments(const float* dets, const int32_t* seg_off, int n_seg, float iou_thr, uint64_t* mask_ws,
                                              uint8_t* keep, void* stream) {
  PT_REQUIRE(seg_off && n_seg >= 1 && n_seg <= NMS_MAXSEG, PT_ELIMIT, "pt_nms_rotated_sorted_segments: 1 <= n_seg <= 16");
  NmsSegs sg{};
  sg.n = n_seg;
  long words = 0;
  int maxn = 0;
  for (int i = 0; i < n_seg; ++i) {
    const int n = seg_off[i + 1] - seg_off[i];
    PT_REQUIRE(n >= 0 && n <= 8192, PT_ELIMIT, "pt_nms_rotated_sorted_segments: at most 8192 candidates per segment");
    sg.off[i] = seg_off[i];
    sg.ws[i] = words;
    words += (long)n * cdiv(n, 64);
    maxn = n > maxn ? n : maxn;
  }
  sg.off[n_seg] = seg_off[n_seg];
  if (maxn == 0) return PT_OK;
  PT_REQUIRE(dets && mask_ws && keep, PT_EINVAL, "pt_nms_rotated_sorted_segments: bad argument");
  const int cols = cdiv(maxn, 64);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(nms_rotated_mask_seg_kernel, dim3(cols, cols, n_seg), dim3(256), 0, s, dets, sg, iou_thr,
                     reinterpret_cast<unsigned long long*>(mask_ws));
  PT_LAUNCH_CHECK("pt_nms_rotated_sorted_segments");
  hipLaunchKernelGGL(nms_scan_seg_kernel, dim3(n_seg), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(mask_ws), sg, keep);
  PT_LAUNCH_CHECK("pt_nms_rotated_sorted_segments");
  return PT_OK;
}

extern "C" int pt_fill_quads_batch(float* img, int B, int C, int H, int W, const float* quads, const uint8_t* alive,
                                   const int32_t* img_of, int Q, float value, void* stream) {
  if (Q == 0) return PT_OK;
  PT_REQUIRE(img && quads && img_of && B > 0 && C > 0 && H > 0 && W > 0 && Q > 0, PT_EINVAL, "pt_fill_quads_batch: bad argument");
  hipLaunchKernelGGL(fill_quads_batch_kernel, dim3(Q), dim3(256), 0, as_stream(stream), img, C, H, W, quads, alive, img_of, value);
  PT_LAUNCH_CHECK("pt_fill_quads_batch");
  return PT_OK;
}
