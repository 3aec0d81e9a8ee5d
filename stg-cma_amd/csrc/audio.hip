// Audio front end of the reference's data loader on the device (AVE/dataloader.py:204-272, `_wav2fbank`): Kaldi-compatible log-mel
// filterbank features -- torchaudio.compliance.kaldi.fbank(htk_compat=True, use_energy=False, window_type='hanning',
// num_mel_bins = 224 | 128, dither = 0, frame_shift = 4.4 | 10 ms; everything else at its default) -- followed by the dataset
// normalisation (x - mean) / (2 std) and the zero padding / cropping to target_length frames, one launch for every 1-second segment of
// a batch.  [S, n] fp32 samples in, [S, target_frames, num_mel_bins] fp32 out: exactly the `a` tensor the models take.
//
// One workgroup = one frame: remove the frame's DC offset, pre-emphasis (replicate-padded), Hann window, power spectrum of the
// zero-padded frame, triangular mel weights, log with the fp32-epsilon floor.  The spectrum is a DIRECT DFT (thread k owns bin k:
// `size` FMAs against a sin / cos table of the padded length in LDS, twiddle index (k j) mod padded): 400 x 257 x 2 FMAs per frame,
// 15 GFMA for a batch of 32 x 10 segments -- under a millisecond, and bit-for-bit independent of an FFT library's butterfly order.
// HBM is not the bound here (16 000 samples in, 224 x 224 values out per segment).
#include <math.h>
#include "common.h"
#include "../../include/stgcma.h"

namespace {

constexpr int FB_THREADS = 256;
constexpr int FB_MAX_PADDED = 2048;      // frames up to 2048 samples after padding (128 ms at 16 kHz)

__global__ void __launch_bounds__(FB_THREADS) fbank_kernel(const float* wave, int64_t n_samples, int64_t wave_stride, int shift, int size,
                                                            int padded, int frames, const float* window, const float* melw, int bins,
                                                            float preemph, float eps, float mean, float inv_std2, int target_frames,
                                                            float* out) {
    extern __shared__ float sm[];                    // frame[padded] | cos[padded] | sin[padded] | power[padded / 2 + 1] | red[8]
    float* fr = sm;
    float* tc = fr + padded;
    float* ts = tc + padded;
    float* pw = ts + padded;
    float* red = pw + padded / 2 + 1;
    const int f = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    float* dst = out + ((int64_t)s * target_frames + f) * bins;
    if (f >= frames) {                               // zero padding behind the last frame (applied AFTER the normalisation, :259-263)
        for (int b = tid; b < bins; b += FB_THREADS) dst[b] = 0.f;
        return;
    }
    const float* x = wave + (int64_t)s * wave_stride + (int64_t)f * shift;
    // twiddles: cos / sin (2 pi i / padded), evaluated in the half-turn form that is exact at the quadrant points
    for (int i = tid; i < padded; i += FB_THREADS) {
        float sv, cv;
        sincospif(2.0f * (float)i / (float)padded, &sv, &cv);
        tc[i] = cv; ts[i] = sv;
    }
    // DC offset of the frame
    float acc = 0.f;
    for (int j = tid; j < size; j += FB_THREADS) acc += x[j];
    acc = wave_sum<64>(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const float dc = (red[0] + red[1] + red[2] + red[3]) / (float)size;
    // pre-emphasis on the DC-free frame (x[-1] := x[0]), window
    for (int j = tid; j < padded; j += FB_THREADS) {
        float v = 0.f;
        if (j < size) {
            const float cur = x[j] - dc, prev = x[j > 0 ? j - 1 : 0] - dc;
            v = (cur - preemph * prev) * window[j];
        }
        fr[j] = v;
    }
    __syncthreads();
    // power spectrum, bins 0 .. padded / 2
    const int nb = padded / 2 + 1;
    for (int k = tid; k < nb; k += FB_THREADS) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int j = 0; j < size; ++j) {
            re = fmaf(fr[j], tc[idx], re);
            im = fmaf(fr[j], ts[idx], im);
            idx += k;
            idx = idx >= padded ? idx - padded : idx;
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    // mel energies, log, normalisation
    for (int b = tid; b < bins; b += FB_THREADS) {
        const float* w = melw + (int64_t)b * nb;
        float e = 0.f;
        for (int k = 0; k < nb; ++k) e = fmaf(pw[k], w[k], e);
        dst[b] = (logf(fmaxf(e, eps)) - mean) * inv_std2;
    }
}

}  // namespace

extern "C" int stg_fbank(const float* wave, int64_t n_samples, int64_t wave_stride, int S, int shift, int size, int padded,
                         const float* window, const float* melw, int num_mel_bins, float preemphasis, float norm_mean,
                         float norm_std, int target_frames, float* out, void* stream) {
    STG_CHECK(wave && window && melw && out, -1, "stg_fbank: null pointer");
    STG_CHECK(S >= 0 && n_samples >= 0 && wave_stride >= n_samples && shift > 0 && size > 0 && padded >= size && padded <= FB_MAX_PADDED &&
              (padded & (padded - 1)) == 0 && num_mel_bins > 0 && target_frames > 0 && norm_std != 0.f, -2,
              "stg_fbank: bad shape (padded: a power of two <= 2048 and >= the frame size)");
    if (S == 0) return 0;
    const int frames = n_samples >= size ? (int)(1 + (n_samples - size) / shift) : 0;       // snip_edges
    const int use = frames < target_frames ? frames : target_frames;                       // frames beyond target_length are cropped
    const size_t lds = (size_t)(3 * padded + padded / 2 + 1 + 8) * sizeof(float);
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)target_frames, (unsigned)S), dim3(FB_THREADS), lds, (hipStream_t)stream, wave, n_samples,
                       wave_stride, shift, size, padded, use, window, melw, num_mel_bins, preemphasis, 1.1920928955078125e-07f, norm_mean,
                       1.0f / (2.0f * norm_std), target_frames, out);
    STG_LAUNCH_CHECK();
    return 0;
}
