// SURVEY 8(f) #1 -- Kaldi-compatible log-mel filterbank + global CMVN on the GPU: the step the reference runs as
//   compute-fbank-feats --config=conf/fbank.conf | apply-cmvn --norm-vars=true data/global_fbank_cmvn.txt
// (AA/infer.py:567-574, AA/conf/fbank.conf:1-4).  Kaldi itself is third-party and absent from the reference tree; the
// algorithm is its published one (feat/feature-window.cc, feat/mel-computations.cc, feat/feature-fbank.cc,
// transform/cmvn.cc) for: 16 kHz, 25 ms / 10 ms frames, snip-edges, no dither, remove-dc-offset, raw log-energy in
// column 0, preemphasis 0.97, hamming window, 512-point FFT, power spectrum, 80 triangular mel bins from 20 Hz to
// Nyquist, log.  PARITY UNPINNED (no Kaldi output to compare with here): tests check it against oracle/oracle.py's
// numpy restatement of the same text.
//
// One wave per frame, four frames per workgroup.  A frame is 400 samples = 1.6 KB: the work is a 512-point FFT in LDS
// (9 radix-2 stages, 4 butterflies per lane and stage, wave-synchronous), 80 short dot products and a row store --
// HBM-bound at 640 B read (with 2.5x overlap from L2) and 324 B written per frame.
#include <math.h>
#include <mutex>
#include <vector>

#include "mdd_internal.h"

namespace mdd {

constexpr int FB_N = 400, FB_S = 160, FB_P = 512, FB_BINS = 80, FB_D = FB_BINS + 1, FB_HALF = FB_P / 2;

struct FbankTables {           // device pointers
    float *window;             // [400]
    float2 *twiddle;           // [256] exp(-2 pi i k / 512)
    int *first, *count, *woff; // [80] first FFT bin, number of bins, offset into weights
    float *weights;
};

#define FB_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__global__ __launch_bounds__(256) void fbank_kernel(const float *__restrict__ wav, int nframes, FbankTables tb,
                                                    const float *__restrict__ cscale, const float *__restrict__ coffset,
                                                    float *__restrict__ out) {
    __shared__ float2 buf[4][FB_P];
    __shared__ float raw[4][FB_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 4 + wave;
    if (f >= nframes) return;                      // wave-uniform: waves never meet at a workgroup barrier
    float2 *z = buf[wave];
    float *xr = raw[wave];
    const float *src = wav + (size_t)f * FB_S;
    // 1. samples, DC offset, raw energy
    float v[7], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int idx = lane + 64 * i;
        v[i] = idx < FB_N ? src[idx] : 0.f;
        sum += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)FB_N;
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int idx = lane + 64 * i;
        if (idx < FB_N) { v[i] -= mean; e += v[i] * v[i]; xr[idx] = v[i]; }
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    const float log_energy = logf(fmaxf(e, 1.1920929e-07f));
    FB_WAVE_SYNC();
    // 2. preemphasis, window, zero padding; stored bit-reversed for the in-place decimation-in-time FFT
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = lane + 64 * i;
        float y = 0.f;
        if (idx < FB_N) y = (v[i < 7 ? i : 0] - 0.97f * xr[idx > 0 ? idx - 1 : 0]) * tb.window[idx];
        z[__brev((unsigned)idx) >> 23] = make_float2(y, 0.f);
    }
    FB_WAVE_SYNC();
    // 3. 9 radix-2 stages
#pragma unroll
    for (int s = 0; s < 9; s++) {
        const int half = 1 << s;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int bfly = lane + 64 * i;                    // 0..255
            const int j = bfly & (half - 1), base = ((bfly >> s) << (s + 1)) + j;
            const float2 w = tb.twiddle[j << (8 - s)];
            const float2 a = z[base], b = z[base + half];
            const float tr = b.x * w.x - b.y * w.y, ti = b.x * w.y + b.y * w.x;
            z[base] = make_float2(a.x + tr, a.y + ti);
            z[base + half] = make_float2(a.x - tr, a.y - ti);
        }
        FB_WAVE_SYNC();
    }
    // 4. power spectrum (bins 0..255 feed the mel banks), kept in the raw buffer
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k = lane + 64 * i;
        const float2 c = z[k];
        xr[k] = c.x * c.x + c.y * c.y;
    }
    FB_WAVE_SYNC();
    // 5. mel energies, log, CMVN, row store
    float *orow = out + (size_t)f * FB_D;
    for (int b = lane; b < FB_BINS; b += 64) {
        const int first = tb.first[b], n = tb.count[b];
        const float *w = tb.weights + tb.woff[b];
        float acc = 0.f;
        for (int k = 0; k < n; k++) acc += w[k] * xr[first + k];
        float val = logf(fmaxf(acc, 1.1920929e-07f));
        if (cscale) val = val * cscale[1 + b] + coffset[1 + b];
        orow[1 + b] = val;
    }
    if (lane == 0) orow[0] = cscale ? log_energy * cscale[0] + coffset[0] : log_energy;
}

static std::mutex g_fb_mutex;
static FbankTables g_fb_tables[16];
static bool g_fb_ready[16];

static int fbank_tables(int dev, FbankTables *out) {
    std::lock_guard<std::mutex> lk(g_fb_mutex);
    if (dev < 0 || dev >= 16) { set_error("mdd_fbank: device %d", dev); return MDD_ERR_ARG; }
    if (!g_fb_ready[dev]) {
        std::vector<float> window(FB_N);
        for (int i = 0; i < FB_N; i++) window[i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / (FB_N - 1)));
        std::vector<float2> tw(FB_HALF);
        for (int k = 0; k < FB_HALF; k++) tw[k] = make_float2((float)cos(-2.0 * M_PI * k / FB_P), (float)sin(-2.0 * M_PI * k / FB_P));
        // MelBanks (no VTLN): triangles in the mel domain, evaluated at the centre frequencies of FFT bins 0..255
        auto mel = [](float fr) { return 1127.0f * logf(1.0f + fr / 700.0f); };
        const float bin_width = 16000.0f / FB_P, mel_low = mel(20.0f), mel_high = mel(8000.0f);
        const float delta = (mel_high - mel_low) / (FB_BINS + 1);
        std::vector<int> first(FB_BINS), count(FB_BINS), woff(FB_BINS);
        std::vector<float> weights;
        for (int b = 0; b < FB_BINS; b++) {
            const float left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
            first[b] = -1; count[b] = 0; woff[b] = (int)weights.size();
            for (int i = 0; i < FB_HALF; i++) {
                const float m = mel(bin_width * i);
                if (m > left && m < right) {
                    if (first[b] < 0) first[b] = i;
                    count[b] = i - first[b] + 1;
                    weights.push_back(m <= center ? (m - left) / (center - left) : (right - m) / (right - center));
                }
            }
            if (first[b] < 0) first[b] = 0;
        }
        FbankTables t;
        MDD_HIP_CHECK(hipMalloc((void **)&t.window, FB_N * sizeof(float)));
        MDD_HIP_CHECK(hipMalloc((void **)&t.twiddle, FB_HALF * sizeof(float2)));
        MDD_HIP_CHECK(hipMalloc((void **)&t.first, FB_BINS * sizeof(int)));
        MDD_HIP_CHECK(hipMalloc((void **)&t.count, FB_BINS * sizeof(int)));
        MDD_HIP_CHECK(hipMalloc((void **)&t.woff, FB_BINS * sizeof(int)));
        MDD_HIP_CHECK(hipMalloc((void **)&t.weights, weights.size() * sizeof(float)));
        MDD_HIP_CHECK(hipMemcpy(t.window, window.data(), FB_N * sizeof(float), hipMemcpyHostToDevice));
        MDD_HIP_CHECK(hipMemcpy(t.twiddle, tw.data(), FB_HALF * sizeof(float2), hipMemcpyHostToDevice));
        MDD_HIP_CHECK(hipMemcpy(t.first, first.data(), FB_BINS * sizeof(int), hipMemcpyHostToDevice));
        MDD_HIP_CHECK(hipMemcpy(t.count, count.data(), FB_BINS * sizeof(int), hipMemcpyHostToDevice));
        MDD_HIP_CHECK(hipMemcpy(t.woff, woff.data(), FB_BINS * sizeof(int), hipMemcpyHostToDevice));
        MDD_HIP_CHECK(hipMemcpy(t.weights, weights.data(), weights.size() * sizeof(float), hipMemcpyHostToDevice));
        g_fb_tables[dev] = t;
        g_fb_ready[dev] = true;
    }
    *out = g_fb_tables[dev];
    return MDD_OK;
}

}  // namespace mdd

using namespace mdd;

extern "C" int32_t mdd_fbank_num_frames(int64_t n_samples) {
    return n_samples < FB_N ? 0 : (int32_t)(1 + (n_samples - FB_N) / FB_S);
}

extern "C" int mdd_fbank(const float *wav_dev, int64_t n_samples, const float *cmvn_scale_dev, const float *cmvn_offset_dev,
                         float *out_dev, void *stream) {
    if (n_samples < 0 || ((cmvn_scale_dev == nullptr) != (cmvn_offset_dev == nullptr))) {
        set_error("mdd_fbank: bad argument"); return MDD_ERR_ARG;
    }
    const int nframes = mdd_fbank_num_frames(n_samples);
    if (nframes == 0) return MDD_OK;                                   // shorter than one window: no rows (snip-edges)
    if (!wav_dev || !out_dev) { set_error("mdd_fbank: null buffer"); return MDD_ERR_ARG; }
    int dev = 0;
    MDD_HIP_CHECK(hipGetDevice(&dev));
    FbankTables tb;
    if (int rc = fbank_tables(dev, &tb)) return rc;
    hipLaunchKernelGGL(fbank_kernel, dim3((nframes + 3) / 4), dim3(256), 0, (hipStream_t)stream, wav_dev, nframes, tb,
                       cmvn_scale_dev, cmvn_offset_dev, out_dev);
    MDD_LAUNCH_CHECK();
    return MDD_OK;
}
