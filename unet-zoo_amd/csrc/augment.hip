// Training-batch assembly on the device: the work of BatchProvider.next_batch + _augmentation_function
// (data/batch_provider.py:43-67,140-271) as ONE gather kernel over a dataset that is resident in HBM (LIDC crops: ~15 k x 128 x 128
// fp32 images + 4 uint8 annotations each = 1.2 GB of the 288 GB).  Per output pixel of sample b:
//   - gather the image / the chosen annotator's label map of dataset row idx[b]                        (:58-63, :131-137)
//   - random crop + resize back to (H, W): cv2.resize(INTER_LINEAR) semantics - source = (o + 0.5) * scale - 0.5, taps
//     clamped to the crop                                                                               (:213-224, utils.py:27-30)
//   - applied to the ROTATED image: cv2.warpAffine(getRotationMatrix2D((W/2, H/2), angle, 1), INTER_LINEAR), constant border 0
//     (dst(x, y) = src(M^-1 (x, y)))                                                                   (:196-208, utils.py:16-20)
//   - labels go through the same two resampling stages as one-hot maps with an argmax after each stage   (utils.py:22-36)
//   - optional left-right / up-down flips                                                               (:253-266)
// The two bilinear stages are evaluated exactly (4 x 4 taps) without materialising the rotated image.  The host draws the random
// parameters with numpy in the reference's order and hands them over as a (B, 8) float table.  OpenCV itself is not in this image:
// its published resampling rules are restated (oracle/augment.py holds the numpy twin); OpenCV's fixed-point coordinate
// quantisation (1/32 pixel in warpAffine) is not reproduced.
#include "uz_common.h"

namespace {

struct AugP {
    const float* X; const uint8_t* Y;      // dataset: images (M, H, W) fp32, labels (M, H, W, A) uint8
    const int* idx; const int* ann;        // per sample: dataset row, annotator
    const float* prm;                      // per sample 8 floats: do_rot, cos, sin, do_scale, p_x, p_y, r, flips (bit 0 lr, bit 1 ud)
    float* xo; float* so;                  // outputs (B, 1, H, W) fp32 image, (B, H, W) fp32 label
    int H, W, A, nlabels;
};

// rotated image value at integer position (x, y): bilinear sample of the source at M^-1 (x, y), zero outside
__device__ __forceinline__ void rot_coords(float c, float s, int W, int H, int x, int y, float& sx, float& sy) {
    // getRotationMatrix2D(center, angle, 1): [a b (1-a) cx - b cy; -b a b cx + (1-a) cy], a = cos, b = sin; warpAffine inverts it
    const float cx = W * 0.5f, cy = H * 0.5f, dx = x - cx, dy = y - cy;
    sx = c * dx - s * dy + cx;
    sy = s * dx + c * dy + cy;
}
__device__ __forceinline__ float img_at(const float* img, int W, int H, int x, int y) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[y * W + x] : 0.f;
}
__device__ __forceinline__ float rotated_image(const float* img, int W, int H, bool rot, float c, float s, int x, int y) {
    if (!rot) return img[y * W + x];
    float sx, sy;
    rot_coords(c, s, W, H, x, y, sx, sy);
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const float fx = sx - x0, fy = sy - y0;
    return (1.f - fy) * ((1.f - fx) * img_at(img, W, H, x0, y0) + fx * img_at(img, W, H, x0 + 1, y0)) +
           fy * ((1.f - fx) * img_at(img, W, H, x0, y0 + 1) + fx * img_at(img, W, H, x0 + 1, y0 + 1));
}
__device__ __forceinline__ int lbl_at(const uint8_t* lab, int W, int H, int A, int x, int y) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? (int)lab[(size_t)(y * W + x) * A] : -1;      // -1: outside (all one-hot channels 0)
}
__device__ __forceinline__ int rotated_label(const uint8_t* lab, int W, int H, int A, int nl, bool rot, float c, float s, int x, int y) {
    if (!rot) return (int)lab[(size_t)(y * W + x) * A];
    float sx, sy;
    rot_coords(c, s, W, H, x, y, sx, sy);
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const float fx = sx - x0, fy = sy - y0;
    const int l00 = lbl_at(lab, W, H, A, x0, y0), l10 = lbl_at(lab, W, H, A, x0 + 1, y0);
    const int l01 = lbl_at(lab, W, H, A, x0, y0 + 1), l11 = lbl_at(lab, W, H, A, x0 + 1, y0 + 1);
    const float w00 = (1.f - fy) * (1.f - fx), w10 = (1.f - fy) * fx, w01 = fy * (1.f - fx), w11 = fy * fx;
    int best = 0; float bv = -1.f;
    for (int k = 0; k < nl; ++k) {              // np.argmax: first maximum wins
        const float v = (l00 == k ? w00 : 0.f) + (l10 == k ? w10 : 0.f) + (l01 == k ? w01 : 0.f) + (l11 == k ? w11 : 0.f);
        if (v > bv) { bv = v; best = k; }
    }
    return best;
}

__global__ __launch_bounds__(256) void augment_k(const AugP p) {
    const int b = blockIdx.y, HW = p.H * p.W;
    const float* q = p.prm + b * 8;
    const bool rot = q[0] != 0.f, sc = q[3] != 0.f;
    const float c = q[1], s = q[2], px = q[4], py = q[5], r = q[6];
    const int flips = (int)q[7];
    const float* img = p.X + (size_t)p.idx[b] * HW;
    const uint8_t* lab = p.Y + (size_t)p.idx[b] * HW * p.A + p.ann[b];
    for (int o = blockIdx.x * 256 + threadIdx.x; o < HW; o += gridDim.x * 256) {
        int oy = o / p.W, ox = o - oy * p.W;
        if (flips & 1) ox = p.W - 1 - ox;
        if (flips & 2) oy = p.H - 1 - oy;
        float v; int l;
        if (!sc) {
            v = rotated_image(img, p.W, p.H, rot, c, s, ox, oy);
            l = rotated_label(lab, p.W, p.H, p.A, p.nlabels, rot, c, s, ox, oy);
        } else {
            // cv2.resize(crop, (W, H), INTER_LINEAR): crop = rotated[py : py + r, px : px + r]; taps clamp to the crop's edge
            const float fxs = (ox + 0.5f) * (r / p.W) - 0.5f, fys = (oy + 0.5f) * (r / p.H) - 0.5f;
            int x0 = (int)floorf(fxs), y0 = (int)floorf(fys);
            float fx = fxs - x0, fy = fys - y0;
            if (x0 < 0) { x0 = 0; fx = 0.f; }
            if (y0 < 0) { y0 = 0; fy = 0.f; }
            int x1 = x0 + 1, y1 = y0 + 1;
            const int rr = (int)r;
            if (x1 > rr - 1) { x1 = rr - 1; if (x0 > rr - 1) x0 = rr - 1; }
            if (y1 > rr - 1) { y1 = rr - 1; if (y0 > rr - 1) y0 = rr - 1; }
            const int X0 = (int)px + x0, X1 = (int)px + x1, Y0 = (int)py + y0, Y1 = (int)py + y1;
            const float w00 = (1.f - fy) * (1.f - fx), w10 = (1.f - fy) * fx, w01 = fy * (1.f - fx), w11 = fy * fx;
            v = w00 * rotated_image(img, p.W, p.H, rot, c, s, X0, Y0) + w10 * rotated_image(img, p.W, p.H, rot, c, s, X1, Y0) +
                w01 * rotated_image(img, p.W, p.H, rot, c, s, X0, Y1) + w11 * rotated_image(img, p.W, p.H, rot, c, s, X1, Y1);
            const int l00 = rotated_label(lab, p.W, p.H, p.A, p.nlabels, rot, c, s, X0, Y0), l10 = rotated_label(lab, p.W, p.H, p.A, p.nlabels, rot, c, s, X1, Y0);
            const int l01 = rotated_label(lab, p.W, p.H, p.A, p.nlabels, rot, c, s, X0, Y1), l11 = rotated_label(lab, p.W, p.H, p.A, p.nlabels, rot, c, s, X1, Y1);
            int best = 0; float bv = -1.f;
            for (int k = 0; k < p.nlabels; ++k) {
                const float t = (l00 == k ? w00 : 0.f) + (l10 == k ? w10 : 0.f) + (l01 == k ? w01 : 0.f) + (l11 == k ? w11 : 0.f);
                if (t > bv) { bv = t; best = k; }
            }
            l = best;
        }
        p.xo[(size_t)b * HW + o] = v;
        p.so[(size_t)b * HW + o] = (float)l;
    }
}

}  // namespace

extern "C" int uz_augment_batch(const float* X, const uint8_t* Y, int H, int W, int A, const int* idx, const int* ann,
                                const float* params, int B, int nlabels, float* x_out, float* s_out, void* stream) {
    UZ_REQUIRE(X && Y && idx && ann && params && x_out && s_out, "augment_batch: null argument");
    UZ_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && A > 0 && nlabels >= 1 && nlabels <= 8, "augment_batch: bad sizes");
    AugP p = {X, Y, idx, ann, params, x_out, s_out, H, W, A, nlabels};
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(augment_k, dim3(gx, B), dim3(256), 0, uz::S(stream), p);
    return uz::check_launch("augment_k");
}
