// k_fft.hip -- generic complex 1024-point Q15 FFT (the cr4_fft_1024_stm32 symbol) and fft() of MFCC.C:27-62 on its own.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_fft_dev.h"

namespace sr {

// ------------------------------------------------------------------------------------------------
// generic complex FFT (cr4_fft_1024_stm32 symbol) and fft() magnitudes: one wave per array
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bitrev8(int v) { return (int)(__brev((uint32_t)v) >> 24); }

// Full-complex passes 1-5, result in natural order in `out` (global).  in/out may not alias in LDS terms.
__device__ void fft_full_wave(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *buf, int lane,
                              const DevTables &t)
{
    // pass 1 (.s:226-232): 256 butterflies, 4 per lane, bit-reversed gather, legs 256 words apart
    // loaded in the order A, C, B, D (.s:134-145); outputs to buf[4*idx + k]
    for (int m = 0; m < 4; m++) {
        const int idx = lane + 64 * m, r = bitrev8(idx);
        const uint32_t wa = in[r], wc = in[r + 256], wb = in[r + 512], wd = in[r + 768];
        int ar = sext_lo(wa), ai = sext_hi(wa), br = sext_lo(wb), bi = sext_hi(wb);
        int cr = sext_lo(wc), ci = sext_hi(wc), dr = sext_lo(wd), di = sext_hi(wd);
        r4_combine<0>(ar, ai, br, bi, cr, ci, dr, di);
        buf[xaddr(4 * idx + 0)] = pack16(ar, ai);
        buf[xaddr(4 * idx + 1)] = pack16(br, bi);
        buf[xaddr(4 * idx + 2)] = pack16(cr, ci);
        buf[xaddr(4 * idx + 3)] = pack16(di, dr);
    }
    wave_sync();
    LaneTw tw;
    load_lane_tw(t, lane, tw);
    uint32_t k2[3][2];
    load_tw3(t, 0, lane & 3, k2);
    const int d0 = lane & 3, d3 = (lane >> 2) & 3, d4 = lane >> 4;
    uint32_t v[4][4], u[4][4];
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
#pragma unroll
        for (int d2 = 0; d2 < 4; d2++) v[d1][d2] = buf[xaddr(d0 + 4 * d1 + 16 * d2 + 64 * d3 + 256 * d4)];
    wave_sync();
#pragma unroll
    for (int d2 = 0; d2 < 4; d2++)
        bfly(v[0][d2], v[1][d2], v[2][d2], v[3][d2], k2[0][0], k2[0][1], k2[1][0], k2[1][1], k2[2][0], k2[2][1]);
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
        bfly(v[d1][0], v[d1][1], v[d1][2], v[d1][3], tw.s3[d1][0][0], tw.s3[d1][0][1], tw.s3[d1][1][0],
             tw.s3[d1][1][1], tw.s3[d1][2][0], tw.s3[d1][2][1]);
    fft_exchange(buf, lane, v, u);
#pragma unroll
    for (int e4 = 0; e4 < 4; e4++)
        bfly(u[0][e4], u[1][e4], u[2][e4], u[3][e4], tw.s4[0][0], tw.s4[0][1], tw.s4[1][0], tw.s4[1][1], tw.s4[2][0],
             tw.s4[2][1]);
#pragma unroll
    for (int e3 = 0; e3 < 4; e3++) {
        bfly(u[e3][0], u[e3][1], u[e3][2], u[e3][3], tw.s5[e3][0][0], tw.s5[e3][0][1], tw.s5[e3][1][0],
             tw.s5[e3][1][1], tw.s5[e3][2][0], tw.s5[e3][2][1]);
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) out[lane + 64 * e3 + 256 * e4] = u[e3][e4];
    }
}

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
