// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path (see sl_oracle.cpp).
//
// CPU restatement of the image preprocessing the reference runs per sample on the host before
// `encode_image` (`foundation_models/clip.py:157-163`: `self.preprocessor(image)`, the transform returned by
// `open_clip.create_model_and_transforms`, `clip.py:88-95`).  The transform itself is third-party and absent from
// /root/reference:
//   * open_clip (pyproject: open-clip-torch) `image_transform(..., is_train=False)`:
//       resize_mode "shortest": Resize(S, BICUBIC) -> CenterCrop(S) -> RGB -> ToTensor -> Normalize(mean, std)
//       resize_mode "squash"  : Resize((S, S), BICUBIC) -> RGB -> ToTensor -> Normalize
//   * torchvision Resize / CenterCrop / ToTensor / Normalize on PIL images (published semantics restated below)
//   * Pillow `Image.resize` (libImaging/Resample.c: antialiased separable convolution, horizontal pass then
//     vertical pass, 8-bit intermediate, 22-bit fixed-point coefficients).
// Parity status: the resize+crop stage is PINNED against Pillow 12.2.0 itself (tests/golden/preprocess.npz, made by
// tests/golden/make_golden_preprocess.py with PIL in the build container); the ToTensor/Normalize stage is two
// IEEE fp32 operations restated from torchvision's documented behaviour (torchvision is not installed here) and
// pinned against torch's own `div`/`sub`/`div` on the same values.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;  // Resample.c PRECISION_BITS

double bicubic_filter(double x) {  // Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

struct Coeffs {
  int ksize;
  std::vector<int> bounds;   // (xmin, count) per output
  std::vector<int32_t> kk;   // ksize fixed-point weights per output
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box (0, inSize)
Coeffs precompute(int inSize, int outSize, int interp) {
  double (*filter)(double) = interp == 0 ? bicubic_filter : bilinear_filter;
  const double fsupport = interp == 0 ? 2.0 : 1.0;
  const float in0 = 0.f, in1 = (float)inSize;
  double scale = (double)(in1 - in0) / outSize, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = fsupport * filterscale;
  Coeffs c;
  c.ksize = (int)std::ceil(support) * 2 + 1;
  c.bounds.resize((size_t)outSize * 2);
  c.kk.assign((size_t)outSize * c.ksize, 0);
  std::vector<double> k(c.ksize);
  for (int xx = 0; xx < outSize; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      const double v = k[x];
      c.kk[(size_t)xx * c.ksize + x] =
          v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    c.bounds[xx * 2] = xmin;
    c.bounds[xx * 2 + 1] = xmax;
  }
  return c;
}

inline uint8_t clip8(int v) {
  v >>= kPrecisionBits;  // arithmetic shift, like the lookup index in Resample.c
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Image.resize((ow, oh), resample) on an 8-bit image with `ch` interleaved channels
std::vector<uint8_t> pil_resize(const uint8_t* in, int h, int w, int ch, int oh, int ow, int interp) {
  if (oh == h && ow == w) return std::vector<uint8_t>(in, in + (size_t)h * w * ch);  // Image.resize: `return self.copy()`
  const bool need_h = ow != w, need_v = oh != h;
  Coeffs ch_ = precompute(w, ow, interp), cv = precompute(h, oh, interp);
  const int ybox_first = cv.bounds[0];
  const int ybox_last = cv.bounds[(size_t)oh * 2 - 2] + cv.bounds[(size_t)oh * 2 - 1];
  std::vector<uint8_t> tmp;
  const uint8_t* src = in;
  int sh = h, sw = w;
  if (need_h) {
    for (int i = 0; i < oh; ++i) cv.bounds[i * 2] -= ybox_first;
    sh = ybox_last - ybox_first;
    sw = ow;
    tmp.resize((size_t)sh * sw * ch);
    for (int yy = 0; yy < sh; ++yy)
      for (int xx = 0; xx < ow; ++xx) {
        const int xmin = ch_.bounds[xx * 2], xmax = ch_.bounds[xx * 2 + 1];
        const int32_t* k = &ch_.kk[(size_t)xx * ch_.ksize];
        for (int c = 0; c < ch; ++c) {
          int ss = 1 << (kPrecisionBits - 1);
          for (int x = 0; x < xmax; ++x) ss += in[((size_t)(yy + ybox_first) * w + x + xmin) * ch + c] * k[x];
          tmp[((size_t)yy * sw + xx) * ch + c] = clip8(ss);
        }
      }
    src = tmp.data();
  }
  if (!need_v) return tmp;
  std::vector<uint8_t> out((size_t)oh * sw * ch);
  for (int yy = 0; yy < oh; ++yy) {
    const int ymin = cv.bounds[yy * 2], ymax = cv.bounds[yy * 2 + 1];
    const int32_t* k = &cv.kk[(size_t)yy * cv.ksize];
    for (int xx = 0; xx < sw; ++xx)
      for (int c = 0; c < ch; ++c) {
        int ss = 1 << (kPrecisionBits - 1);
        for (int y = 0; y < ymax; ++y) ss += src[((size_t)(y + ymin) * sw + xx) * ch + c] * k[y];
        out[((size_t)yy * sw + xx) * ch + c] = clip8(ss);
      }
  }
  (void)sh;
  return out;
}

}  // namespace

// torchvision.transforms.functional._compute_resized_output_size for an int `size` (shortest edge -> S,
// the other edge int(S * long / short)); "squash" gives (S, S).
ORC_API void orc_resized_size(int h, int w, int S, int resize_mode, int* oh, int* ow) {
  if (resize_mode == 1) {
    *oh = S;
    *ow = S;
    return;
  }
  const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
  const int new_long = (int)((double)((int64_t)S * lng) / (double)shrt);
  if (w <= h) {
    *ow = S;
    *oh = new_long;
  } else {
    *oh = S;
    *ow = new_long;
  }
}

// torchvision CenterCrop offset: int(round((size - crop) / 2.0)), Python round = half to even
ORC_API int orc_center_crop_offset(int size, int crop) {
  const int d = size - crop;
  const int fl = (int)std::floor(d / 2.0);
  if ((d & 1) == 0) return fl;
  return (fl & 1) ? fl + 1 : fl;
}

// pixels: (h, w, 3) uint8 RGB.  out_u8 (optional): (S, S, 3) resized+cropped bytes; out_f32: (3, S, S) normalised.
ORC_API int orc_preprocess(const uint8_t* pixels, int h, int w, int S, int resize_mode, int interp, const float* mean,
                           const float* stdv, uint8_t* out_u8, float* out_f32) {
  int oh, ow;
  orc_resized_size(h, w, S, resize_mode, &oh, &ow);
  if (oh < S || ow < S) return -1;  // CenterCrop would pad; not produced by Resize(S) + CenterCrop(S)
  std::vector<uint8_t> r = pil_resize(pixels, h, w, 3, oh, ow, interp);
  const int top = orc_center_crop_offset(oh, S), left = orc_center_crop_offset(ow, S);
  for (int y = 0; y < S; ++y)
    for (int x = 0; x < S; ++x)
      for (int c = 0; c < 3; ++c) {
        const uint8_t u = r[((size_t)(y + top) * ow + x + left) * 3 + c];
        if (out_u8) out_u8[((size_t)y * S + x) * 3 + c] = u;
        // ToTensor: float(u) / 255 ; Normalize: (t - mean) / std, each one IEEE fp32 operation
        volatile float t = (float)u / 255.0f;
        volatile float d = t - mean[c];
        if (out_f32) out_f32[((size_t)c * S + y) * S + x] = d / stdv[c];
      }
  return 0;
}
