// OpenEarthMap tile preparation on the GPU: everything dataset/base_dataset.py does to a decoded tile between
// rasterio's read and the training loop, fused into one pass per batch --
//   crop (:140-174) -> pad to the crop size, image 0 / label ignore (:88-104) -> horizontal flip (:106-110) -> rot90 x k (:134-138)
//   -> channel reversal + /255 + (x - mean) / std (:29-34) -> CHW float image, int64 label (:36-43),
// plus the label re-indexing of dataset/oem.py:113-133 / oem_ft.py:197 as a 256-entry lookup table.  The random draws (crop offsets, flip,
// k) stay on the host, in the reference's order; decoding the GeoTIFF stays on the host too (rasterio).  One thread per output pixel.
#include "common.h"

namespace {

struct AugTile {            // 32 bytes, device table entry
  const uint8_t* img;       // [H][W][3] uint8 (rasterio (3,H,W) -> np.rollaxis(image, 0, 3))
  const uint8_t* lbl;       // [H][W] uint8, or null (unlabeled test tiles)
  int H, W;
  int h_off, w_off;         // crop origin
};
struct AugFlags { int flip, rot_k; };

__global__ __launch_bounds__(256) void augment_kernel(const AugTile* __restrict__ tiles, const int* __restrict__ flags, int B, int ch, int cw,
                                                      double m0, double m1, double m2, double s0, double s1, double s2, int ignore_label,
                                                      const uint8_t* __restrict__ lut, float* __restrict__ out_img, long long* __restrict__ out_lbl) {
  const long long total = (long long)B * ch * cw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw), y = (int)((i / cw) % ch), b = (int)(i / ((long long)cw * ch));
    const AugTile t = tiles[b];
    const int flip = flags[2 * b], k = flags[2 * b + 1] & 3;
    // np.rot90(m, k, (0, 1)) on the square crop: out[y][x] = m[r][c]
    int r, c;
    if (k == 0) { r = y; c = x; }
    else if (k == 1) { r = x; c = cw - 1 - y; }
    else if (k == 2) { r = ch - 1 - y; c = cw - 1 - x; }
    else { r = ch - 1 - x; c = y; }
    if (flip) c = cw - 1 - c;                       // np.flip(axis=1) happened before the rotation
    const int sy = t.h_off + r, sx = t.w_off + c;   // crop; beyond the tile: the padding
    const bool inside = r < t.H - t.h_off && c < t.W - t.w_off && sy < t.H && sx < t.W;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    long long lab = ignore_label;
    if (inside) {
      const uint8_t* p = t.img + ((size_t)sy * t.W + sx) * 3;
      v0 = (float)p[2]; v1 = (float)p[1]; v2 = (float)p[0];      // image[:, :, ::-1]
      if (t.lbl) { const int l = t.lbl[(size_t)sy * t.W + sx]; lab = lut ? lut[l] : l; }
    }
    const size_t plane = (size_t)ch * cw, o = (size_t)b * 3 * plane + (size_t)y * cw + x;
    // numpy: float32 / 255.0 stays float32; `image -= mean` / `image /= std` with python-list operands compute in float64 and store float32
    out_img[o] = (float)((double)(float)((double)(v0 / 255.0f) - m0) / s0);
    out_img[o + plane] = (float)((double)(float)((double)(v1 / 255.0f) - m1) / s1);
    out_img[o + 2 * plane] = (float)((double)(float)((double)(v2 / 255.0f) - m2) / s2);
    if (out_lbl) out_lbl[(size_t)b * plane + (size_t)y * cw + x] = lab;
  }
}

}  // namespace

extern "C" int sl_augment_batch(const void* tiles_dev, const int* flip_rot_dev, int B, int crop_h, int crop_w, const double* mean3_host,
                                const double* std3_host, int ignore_label, const uint8_t* label_lut_dev, float* out_img, long long* out_lbl,
                                sl_stream_t stream) {
  SL_REQUIRE(tiles_dev && flip_rot_dev && out_img && mean3_host && std3_host && B > 0 && crop_h > 0 && crop_w > 0, "augment_batch: bad args");
  static_assert(sizeof(AugTile) == 32, "table layout is part of the ABI");
  const long long n = (long long)B * crop_h * crop_w;
  const int grid = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  hipLaunchKernelGGL(augment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const AugTile*)tiles_dev, flip_rot_dev, B, crop_h, crop_w,
                     mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], ignore_label, label_lut_dev, out_img, out_lbl);
  SL_LAUNCH_CHECK("augment_kernel");
  return 0;
}
