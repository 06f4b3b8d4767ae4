// k_datas.hip -- the training input pipeline on the device.
//
//   datas/us1k.py:16-36 crop_patch: LR corner (lx, ly), HR corner (lx*s, ly*s); optional [:, ::-1], [::-1, :],
//   transpose(1,0,2); utils.ndarray2tensor (HWC -> CHW, .float()); datas/us1k.py:169: / 255.
//
// MI355X-first: the whole npy cache (uint8 HWC images) stays resident in HBM -- 288 GB holds it many times over --
// and a batch is cut straight out of it: no worker processes, no pinned staging, no H2D copy per step.  The random
// draws stay on the host (Python's `random`, the reference's order) and arrive as a small descriptor table in the
// kernel arguments; the kernel is pure index arithmetic plus one correctly-rounded fp32 division, so its output is
// bit-identical to the reference's tensors.
#include "m2t_common.h"
#include "m2t_kernels.h"
#include "../../include/m2t.h"

namespace {

constexpr int DESC_PER_LAUNCH = 32;

struct PatchDesc {
  long long lr_off, hr_off;      // byte offset of the image in its pool
  int lr_w, hr_w;                // row length of the image in pixels
  int lx, ly;                    // LR corner (datas/us1k.py:21)
  int flags, pad;                // bit0 hflip, bit1 vflip, bit2 transpose
};
struct PatchTable { PatchDesc d[DESC_PER_LAUNCH]; };

// grid (ceil((lp^2 + hp^2) / 256), samples of this launch); one thread per output pixel, all channels
__global__ __launch_bounds__(256) void crop_patches_kernel(const uint8_t* __restrict__ lr_pool, const uint8_t* __restrict__ hr_pool,
                                                           PatchTable tab, int first, int C, int lp, int scale,
                                                           float* __restrict__ lr_out, float* __restrict__ hr_out) {
  const PatchDesc d = tab.d[blockIdx.y];
  const int hp = lp * scale, nl = lp * lp, nh = hp * hp;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nl + nh) return;
  const bool is_hr = i >= nl;
  if (is_hr) i -= nl;
  const int n = is_hr ? hp : lp, w = is_hr ? d.hr_w : d.lr_w;
  int y = i / n, x = i - y * n;                       // output pixel
  if (d.flags & 4) { const int t = y; y = x; x = t; } // p3[y][x] = p2[x][y]
  if (d.flags & 2) y = n - 1 - y;                     // p2[y][x] = p1[n-1-y][x]
  if (d.flags & 1) x = n - 1 - x;                     // p1[y][x] = p0[y][n-1-x]
  const int cy = (is_hr ? d.ly * scale : d.ly) + y, cx = (is_hr ? d.lx * scale : d.lx) + x;
  const uint8_t* src = (is_hr ? hr_pool + d.hr_off : lr_pool + d.lr_off) + ((long long)cy * w + cx) * C;
  float* dst = (is_hr ? hr_out : lr_out) + (long long)(first + blockIdx.y) * C * (is_hr ? nh : nl) + i;
  for (int c = 0; c < C; ++c) dst[(long long)c * (is_hr ? nh : nl)] = __fdiv_rn((float)src[c], 255.f);
}

// whole image (datas/benchmark.py:62-72): top-left h x w crop of a uint8 HWC image -> float32 CHW / 255
__global__ __launch_bounds__(256) void image_to_tensor_kernel(const uint8_t* __restrict__ img, int img_w, int C, int h, int w,
                                                              float* __restrict__ out) {
  const long long n = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (long long)y * w);
    const uint8_t* src = img + ((long long)y * img_w + x) * C;
    for (int c = 0; c < C; ++c) out[(long long)c * n + i] = __fdiv_rn((float)src[c], 255.f);
  }
}

}  // namespace

extern "C" int m2t_crop_patches(const unsigned char* lr_pool, const unsigned char* hr_pool, const long long* desc_host, int n,
                                int channels, int patch_size, int scale, float* lr_out, float* hr_out, void* stream) {
  if (!lr_pool || !hr_pool || !desc_host || !lr_out || !hr_out || n < 1 || channels < 1 || scale < 1 || patch_size < scale)
    return m2t_set_error(M2T_ERR_ARG, "m2t_crop_patches: bad argument");
  const int lp = patch_size / scale;                  // datas/us1k.py:20
  if (lp * scale != patch_size) return m2t_set_error(M2T_ERR_ARG, "m2t_crop_patches: patch_size must be a multiple of scale");
  for (int i = 0; i < n; ++i) {
    const long long* q = desc_host + 8LL * i;
    if (q[0] < 0 || q[1] < 0 || q[4] < 0 || q[5] < 0 || q[4] + lp > q[2] || (q[4] + lp) * scale > q[3] || q[5] + lp > q[7] || (q[6] & ~7LL))
      return m2t_set_error(M2T_ERR_ARG, "m2t_crop_patches: descriptor out of range");
  }
  const int total = lp * lp + patch_size * patch_size;
  for (int first = 0; first < n; first += DESC_PER_LAUNCH) {
    const int m = n - first < DESC_PER_LAUNCH ? n - first : DESC_PER_LAUNCH;
    PatchTable tab;
    for (int i = 0; i < m; ++i) {
      const long long* q = desc_host + 8LL * (first + i);
      tab.d[i] = PatchDesc{q[0], q[1], (int)q[2], (int)q[3], (int)q[4], (int)q[5], (int)q[6], 0};
    }
    for (int i = m; i < DESC_PER_LAUNCH; ++i) tab.d[i] = tab.d[0];
    crop_patches_kernel<<<dim3((total + 255) / 256, m), 256, 0, (hipStream_t)stream>>>(lr_pool, hr_pool, tab, first, channels, lp, scale,
                                                                                      lr_out, hr_out);
    M2T_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int m2t_image_to_tensor(const unsigned char* img, int img_h, int img_w, int channels, int h, int w, float* out, void* stream) {
  if (!img || !out || channels < 1 || h < 1 || w < 1 || h > img_h || w > img_w)
    return m2t_set_error(M2T_ERR_ARG, "m2t_image_to_tensor: bad argument (crop must fit the image)");
  const long long n = (long long)h * w;
  const int g = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  image_to_tensor_kernel<<<g, 256, 0, (hipStream_t)stream>>>(img, img_w, channels, h, w, out);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// On-device batch augmentations of train.py:177-181 (utils.py:16-108): cutmix and cut_out as ONE box-table kernel.
// The random draws stay with the host (reference order, m2trans_amd/augment.py); the table holds per sample up to
// `maxb` boxes (x1, y1, x2, y2, source sample) in the order the reference applies them:
//   mode 0 (cutmix, utils.py:36-51): a pixel takes the value of the LAST box that covers it, read from the source sample of
//          that box in the ORIGINAL tensor (every patch copies from the original, later patches overwrite);
//   mode 1 (cut_out, utils.py:74-92): a pixel covered by any box is multiplied by 0 (img * mask).
// `mult` scales the box (1 for the LR tensor, `scale` for the HR tensor, utils.py:49).
// =======================================================================================
__global__ void __launch_bounds__(256) box_mix_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W,
                                                      const int* __restrict__ table, int maxb, int mode, int mult) {
  const long long total = (long long)B * C * H * W;
  const int rowlen = 1 + 5 * maxb;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % W);
    long long r = t / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const int* row = table + (long long)b * rowlen;
    const int n = row[0];
    float v = src[t];
    for (int i = n - 1; i >= 0; --i) {
      const int* q = row + 1 + 5 * i;
      if (x >= q[0] * mult && x < q[2] * mult && y >= q[1] * mult && y < q[3] * mult) {
        v = (mode == 0) ? src[(((long long)q[4] * C + c) * H + y) * W + x] : v * 0.0f;
        break;
      }
    }
    dst[t] = v;
  }
}
extern "C" int m2t_box_mix(const float* src, float* dst, int B, int C, int H, int W, const int* table_dev, int max_boxes, int mode,
                           int mult, void* stream) {
  if (!src || !dst || !table_dev || B < 1 || C < 1 || H < 1 || W < 1 || max_boxes < 1 || mult < 1 || (mode != 0 && mode != 1))
    return m2t_set_error(M2T_ERR_ARG, "m2t_box_mix: bad argument");
  if (src == dst) return m2t_set_error(M2T_ERR_ARG, "m2t_box_mix: in-place is not supported (boxes read other samples of the original)");
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(box_mix_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     B, C, H, W, table_dev, max_boxes, mode, mult);
  M2T_LAUNCH_CHECK();
  return 0;
}
