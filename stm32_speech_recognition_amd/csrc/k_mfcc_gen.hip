// k_mfcc_gen.hip -- get_mfcc (MFCC.C:86-191) for the GENERIC front end: any sampling rate and framing the VAD kernel is
// built for (frame_len <= 1024), the reference's 1024-point Q15 transform, 4..64 Mel filters (even), 1..16 coefficients.
// The reference's constants are compile-time (#defines in MFCC.H:7-16, VAD.H:4-8, pasted tables in MFCC_Arg.h); here they
// are run-time values with the tables generated from the same Matlab formulas (csrc/sr_tables.cpp), and the arithmetic is
// the reference's rule for rule, checked bit for bit against the parametrised oracle.  No reference counterpart for the
// constants themselves.  A straightforward kernel (one wave per frame, every stage through LDS, the generic complex
// transform of k_fft.hip): roughly 4x slower per frame than the two specialised kernels, which stay untouched.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; integer VALU + LDS.
#include "sr_fft_dev.h"

namespace sr {

constexpr int kGenWaves = 4;
constexpr int kGenMaxMel = 64;  // sr_create: n_mel <= 64, n_coef <= 16 (one lane per filter / coefficient)
// per-wave LDS words: transform input, output, exchange scratch, bin energies, filterbank outputs
constexpr int kGenWaveWords = kNfft + kNfft + kXchgWords + kBins + kGenMaxMel;

__global__ void __launch_bounds__(64 * kGenWaves) k_mfcc_gen(const MfccArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *fin = smem + (size_t)w * kGenWaveWords, *fout = fin + kNfft, *buf = fout + kNfft, *en = buf + kXchgWords,
             *pw = en + kBins;
    const uint32_t FL = a.frame_len, hop = a.hop, nm = a.n_mel, nc = a.n_coef;
    const uint64_t n_items = (uint64_t)a.B * a.max_frames;
    for (uint64_t item = (uint64_t)blockIdx.x * kGenWaves + w; item < n_items; item += (uint64_t)gridDim.x * kGenWaves) {
        const uint32_t b = (uint32_t)(item / a.max_frames), f = (uint32_t)(item - (uint64_t)b * a.max_frames);
        const sr_vad_rec *rec = a.vad + b;
        int16_t *out = a.mfcc + ((uint64_t)b * a.max_frames + f) * nc;
        if (f >= rec->frm_num) {  // rows >= frm_num are zero so that every row of the output is defined
            if ((uint32_t)lane < nc) out[lane] = 0;
            continue;
        }
        const int mid = (int)rec->atap.mid_val;
        const uint16_t *x = a.pcm + (uint64_t)b * a.pcm_stride + rec->seg[0] + (uint64_t)hop * f;
        // ---- pre-emphasis + Hamming (MFCC.C:115-124); x[-1] is the sample before the frame; zero padding to 1024
        for (uint32_t i = lane; i < (uint32_t)kNfft; i += 64) {
            uint32_t v = 0;
            if (i < FL) {
                const int cur = (int)x[i] - mid, prv = (int)x[(int)i - 1] - mid;
                const int t = cur - preemph95(prv);
                v = (uint32_t)(uint16_t)(int16_t)(mul24(t, (int)a.t.hamm[i]) / 1000);  // real part, imaginary half 0
            }
            fin[i] = v;
        }
        wave_sync();
        fft_full_wave(fin, fout, buf, lane, a.t);  // cr4_fft_1024_stm32 (.s:95-281)
        wave_sync();
        // ---- |X| * 10 and energy (MFCC.C:49-60, 128-133), u32 wrap
        for (int k = lane; k < kBins; k += 64) {
            const uint32_t wd = fout[k];
            const uint32_t m = cvt_u32(sqrt_rn_int((float)sdot2z(wd, wd)) * 10.0f);
            en[k] = m * m;
        }
        wave_sync();
        // ---- Mel filterbank (MFCC.C:136-162): filter h sums E*tri/100 term by term (u32 wrap) over bins [lo, hi) of the
        //      even / odd poly-line; then the log (MFCC.C:165-170)
        if ((uint32_t)lane < nm) {
            const uint32_t h = (uint32_t)lane;
            const uint32_t lo = (h == 0) ? 0u : a.t.tri_cen[h - 1], hi = (h == nm - 1) ? (uint32_t)kBins : a.t.tri_cen[h + 1];
            const uint16_t *tri = (h & 1) ? a.t.tri_odd : a.t.tri_even;
            uint32_t acc = 0;
            for (uint32_t k = lo; k < hi; k++) acc += en[k] * (uint32_t)tri[k] / 100u;
            pw[h] = log100_u32(acc, a.t.log_thr);
        }
        wave_sync();
        // ---- DCT (MFCC.C:173-183): per-term truncating / 100, s16 accumulator (= the low 16 bits of the integer sum)
        if ((uint32_t)lane < nc) {
            const int8_t *d = a.t.dct + (uint32_t)lane * nm;
            int acc = 0;
            for (uint32_t h = 0; h < nm; h++) acc += (int)(int16_t)((int)pw[h] * (int)d[h] / 100);
            out[lane] = (int16_t)acc;
        }
        wave_sync();
    }
}

void launch_mfcc_gen(const MfccArgs &a, hipStream_t s)
{
    const uint64_t n = (uint64_t)a.B * a.max_frames;
    if (!n) return;
    const uint64_t wgs = (n + kGenWaves - 1) / kGenWaves;
    const uint32_t grid = (uint32_t)(wgs < 8192 ? wgs : 8192);
    hipLaunchKernelGGL(k_mfcc_gen, dim3(grid), dim3(64 * kGenWaves), (size_t)kGenWaves * kGenWaveWords * sizeof(uint32_t), s, a);
}

}  // namespace sr
