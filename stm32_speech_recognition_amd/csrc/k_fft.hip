// k_fft.hip -- generic complex 1024-point Q15 FFT (the cr4_fft_1024_stm32 symbol) and fft() of MFCC.C:27-62 on its own.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_fft_dev.h"

namespace sr {

// ------------------------------------------------------------------------------------------------
// generic complex FFT (cr4_fft_1024_stm32 symbol) and fft() magnitudes: one wave per array
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fft_q15(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables t)
{
    __shared__ uint32_t buf[kXchgWords];
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        fft_full_wave(in + (size_t)i * kNfft, out + (size_t)i * kNfft, buf, lane, t);
        wave_sync();
    }
}

void launch_fft_q15(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_fft_q15, dim3(n < 4096 ? n : 4096), dim3(64), 0, s, in, out, n, t);
}

// fft() of MFCC.C:27-62 on its own: zero-extend len samples, FFT, magnitudes of bins 0..511.
__global__ void __launch_bounds__(64) k_fft_mag(const int16_t *frames, uint32_t len, uint32_t *mag, uint32_t *raw_hi,
                                                uint32_t n, const DevTables t)
{
    __shared__ uint32_t buf[kXchgWords];
    __shared__ uint32_t fin[kNfft], fout[kNfft];
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        for (int k = lane; k < kNfft; k += 64) fin[k] = (k < (int)len) ? (uint32_t)(uint16_t)frames[(size_t)i * len + k] : 0u;
        wave_sync();
        fft_full_wave(fin, fout, buf, lane, t);
        wave_sync();
        for (int k = lane; k < kBins; k += 64) {
            const int re = sext_lo(fout[k]), im = sext_hi(fout[k]);
            const int r = re * re + im * im;
            mag[(size_t)i * kBins + k] = (uint32_t)(sqrtf((float)r) * 10.0f);
            if (raw_hi) raw_hi[(size_t)i * kBins + k] = fout[kBins + k];
        }
        wave_sync();
    }
}

void launch_fft_mag(const int16_t *frames, uint32_t len, uint32_t *mag, uint32_t *raw_hi, uint32_t n, const DevTables &t,
                    hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_fft_mag, dim3(n < 4096 ? n : 4096), dim3(64), 0, s, frames, len, mag, raw_hi, n, t);
}

}  // namespace sr
