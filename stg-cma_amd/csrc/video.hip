// Video front end on the device (video half): the TENSOR part of the reference's training-time frame pipeline,
// AVE/dataloader.py:346-394 (_aug_frame_train behind the PIL RandAugment), for frames that are already decoded, in ONE launch per batch:
//   ToTensor (u8 / 255)  ->  tensor_normalize ((x - mean) / std, :470-485)  ->  random_resized_crop (:442-452; transforms/video_transforms.py:529-561:
//   crop box (i, j, h, w), torch.nn.functional.interpolate(bilinear, align_corners=False) to S x S)  ->  horizontal_flip (:453-454)  ->
//   RandomErasing 'pixel' mode, one box per clip, fresh normal noise per frame (transforms/random_erasing.py:118-152)  ->  'C T H W'.
// The random draws (crop box, flip, erase box) are ARGUMENTS -- one int32[9] record per clip -- and so is the erase noise: the reference
// draws them from Python's / NumPy's / torch's host generators, whose streams a device kernel cannot reproduce; the caller draws them
// (stg-cma_amd/video.py does, with the reference's distributions).  File decoding and the PIL RandAugment stay on the host.
//
// HBM-bound, trivially: per clip 3 T H W bytes in (the crop's rows only), 12 T S^2 bytes out.  One thread per output pixel (t, y, x), the
// three channels of a tap are one 3-byte read of the HWC frame; consecutive threads write consecutive x of each channel plane.
#include "common.h"
#include "../../include/stgcma.h"

namespace {

struct VidP {
    const uint8_t* frames;      // [B][T][H][W][3]
    const int32_t* params;      // [B][9]: i, j, h, w, flip, erase top, left, height, width (in output coordinates; height 0 = no erase)
    const float* noise;         // [B][T][3][S][S] (read inside the erase box only) or null
    float* out;                 // [B][3][T][S][S]
    int B, T, H, W, S;
    float mean[3], istd_dummy[3], stdv[3];
};

// PyTorch's upsample_bilinear2d source index (align_corners = False): src = scale * (dst + 0.5) - 0.5, clamped at 0
__device__ __forceinline__ void src_index(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

__global__ void __launch_bounds__(256) video_aug_kernel(VidP p) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.S * p.S) return;
    const int y = idx / p.S, x = idx - y * p.S;
    const int32_t* q = p.params + b * 9;
    const int ci = q[0], cj = q[1], ch = q[2], cw = q[3], flip = q[4], et = q[5], el = q[6], eh = q[7], ew = q[8];
    const int64_t plane = (int64_t)p.S * p.S;
    float* o = p.out + ((int64_t)b * 3 * p.T + t) * plane + idx;                 // channel c at + c * T * plane
    if (eh > 0 && y >= et && y < et + eh && x >= el && x < el + ew && p.noise) {
        const float* n = p.noise + ((int64_t)(b * p.T + t) * 3) * plane + idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(int64_t)c * p.T * plane] = n[c * plane];
        return;
    }
    const int xs = flip ? p.S - 1 - x : x;                                       // the flip follows the resize: mirror the output column
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(y, (float)ch / (float)p.S, ch, y0, y1, ly0, ly1);
    src_index(xs, (float)cw / (float)p.S, cw, x0, x1, lx0, lx1);
    const uint8_t* f = p.frames + ((int64_t)(b * p.T + t) * p.H) * p.W * 3;
    const uint8_t* r0 = f + ((int64_t)(ci + y0) * p.W + cj) * 3;
    const uint8_t* r1 = f + ((int64_t)(ci + y1) * p.W + cj) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // normalise every tap like the reference does before it resizes: (u8 / 255 - mean) / std, true divisions
        const float v00 = ((float)r0[x0 * 3 + c] / 255.0f - p.mean[c]) / p.stdv[c];
        const float v01 = ((float)r0[x1 * 3 + c] / 255.0f - p.mean[c]) / p.stdv[c];
        const float v10 = ((float)r1[x0 * 3 + c] / 255.0f - p.mean[c]) / p.stdv[c];
        const float v11 = ((float)r1[x1 * 3 + c] / 255.0f - p.mean[c]) / p.stdv[c];
        o[(int64_t)c * p.T * plane] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
}

}  // namespace

extern "C" int stg_video_aug(const void* frames, int B, int T, int H, int W, const int32_t* params, const float* noise,
                             const float* mean3, const float* std3, float* out, int S, void* stream) {
    STG_CHECK(frames && params && mean3 && std3 && out, -1, "stg_video_aug: null pointer");
    STG_CHECK(B > 0 && T > 0 && H > 0 && W > 0 && S > 0 && B <= 65535 && T <= 65535, -2, "stg_video_aug: bad shape");
    VidP p;
    p.frames = (const uint8_t*)frames; p.params = params; p.noise = noise; p.out = out;
    p.B = B; p.T = T; p.H = H; p.W = W; p.S = S;
    for (int c = 0; c < 3; ++c) {
        STG_CHECK(std3[c] != 0.f, -2, "stg_video_aug: zero std");
        p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; p.istd_dummy[c] = 0.f;
    }
    hipLaunchKernelGGL(video_aug_kernel, dim3((unsigned)((S * S + 255) / 256), (unsigned)T, (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
    STG_LAUNCH_CHECK();
    return 0;
}
