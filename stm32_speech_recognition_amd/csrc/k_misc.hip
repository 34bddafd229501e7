// k_misc.hip -- delta cepstra (EXTENSION), and the scalar diagnostics / compat helpers: log / sqrt device functions swept directly, get_dis, dtw_limit.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_dtw_dev.h"

namespace sr {

// ------------------------------------------------------------------------------------------------
// k_delta_mfcc: EXTENSION, no reference counterpart (the accompanying thesis, p.32, lists difference cepstra as future
// work).  Two-frame regression over the s16 MFCC rows of a record, d[t] = ((m[t+1]-m[t-1]) + 2(m[t+2]-m[t-2])) / 10 with
// rows clamped to [0, n-1], s32 arithmetic, division truncating toward zero (as every division of MFCC.C); rows >= n
// are zero.  Defined in oracle/sr_oracle.c (sr_oracle_delta_mfcc); pure streaming: one thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_delta_mfcc(const int16_t *mfcc, const sr_vad_rec *vad, const uint32_t *frames,
                                                    uint32_t B, uint32_t max_frames, uint32_t nc, int16_t *delta)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = max_frames * nc;
    if (i >= (uint64_t)B * per) return;
    const uint32_t b = (uint32_t)(i / per), r = (uint32_t)(i - (uint64_t)b * per), t = r / nc, c = r - t * nc;
    uint32_t n = frames ? frames[b] : ((vad[b].status == SR_ST_OK) ? vad[b].frm_num : 0u);
    if (n > max_frames) n = max_frames;
    int16_t out = 0;
    if (t < n) {
        const int16_t *m = mfcc + (uint64_t)b * per + c;
        const uint32_t p1 = t + 1 < n ? t + 1 : n - 1, p2 = t + 2 < n ? t + 2 : n - 1;
        const uint32_t m1 = t >= 1 ? t - 1 : 0, m2 = t >= 2 ? t - 2 : 0;
        const int num = ((int)m[p1 * nc] - (int)m[m1 * nc]) + 2 * ((int)m[p2 * nc] - (int)m[m2 * nc]);
        out = (int16_t)(num / 10);
    }
    delta[i] = out;
}
void launch_delta_mfcc(const int16_t *mfcc, const sr_vad_rec *vad, const uint32_t *frames, uint32_t B, uint32_t max_frames,
                       uint32_t n_coef, int16_t *delta, hipStream_t s)
{
    const uint64_t n = (uint64_t)B * max_frames * n_coef;
    if (!n) return;
    hipLaunchKernelGGL(k_delta_mfcc, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, mfcc, vad, frames, B, max_frames, n_coef, delta);
}

// ------------------------------------------------------------------------------------------------
// k_unpack12: host transport of 12-bit ADC codes (ADC.H:7-11: the STM32's 12-bit converter) packed two samples in three
// bytes -- sample 2i = b[3i] | (b[3i+1] & 0x0F) << 8, sample 2i+1 = b[3i+1] >> 4 | b[3i+2] << 4 -- back into the u16
// capture rows every kernel of the path reads.  One thread per 8 samples: 12 bytes in (three aligned dwords), 16 bytes out.
// A streaming kernel (HBM-bound: 1.5 + 2 bytes per sample); it exists so that the PCIe upload moves 25 % fewer bytes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack12(const uint32_t *packed, uint64_t row_words, uint16_t *out, uint64_t out_stride,
                                                  uint32_t groups_per_row, uint32_t B)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)B * groups_per_row) return;
    const uint32_t b = (uint32_t)(i / groups_per_row), g = (uint32_t)(i - (uint64_t)b * groups_per_row);
    const uint32_t *src = packed + (uint64_t)b * row_words + 3ull * g;
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];  // 24 nibbles, 8 samples of 3 nibbles each, little-endian
    const uint32_t s0 = w0 & 0xFFFu, s1 = (w0 >> 12) & 0xFFFu, s2 = (w0 >> 24) | ((w1 & 0xFu) << 8), s3 = (w1 >> 4) & 0xFFFu;
    const uint32_t s4 = (w1 >> 16) & 0xFFFu, s5 = (w1 >> 28) | ((w2 & 0xFFu) << 4), s6 = (w2 >> 8) & 0xFFFu, s7 = w2 >> 20;
    *(uint4 *)(out + (uint64_t)b * out_stride + 8ull * g) = make_uint4(s0 | s1 << 16, s2 | s3 << 16, s4 | s5 << 16, s6 | s7 << 16);
}
void launch_unpack12(const void *packed, uint64_t row_bytes, uint16_t *out, uint64_t out_stride, uint32_t buf_len, uint32_t B,
                     hipStream_t s)
{
    const uint32_t groups = (buf_len + 7) / 8;
    const uint64_t n = (uint64_t)B * groups;
    if (!n) return;
    hipLaunchKernelGGL(k_unpack12, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t *)packed, row_bytes / 4, out,
                       out_stride, groups, B);
}

// ------------------------------------------------------------------------------------------------
// diagnostics: the three non-integer device functions on their own, so tests can sweep them directly
//   out[3i+0] = (u32)(log((double)x)*100)                       MFCC.C:168   (step-function evaluation)
//   out[3i+1] = (u32)sqrtf((float)x)                            DTW.C:59     (v_rsq_f32 seed + fused correction, sqrt_rn_int)
//   out[3i+2] = (u32)(sqrtf((float)(s32)x)*10), x < 2^31        MFCC.C:56-58
// ------------------------------------------------------------------------------------------------
__global__ void k_math_diag(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *log_thr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = in[i];
    out[3 * i + 0] = log100_u32(x, log_thr);
    // the exact routine and the from-below root of the DTW kernel's step are reported through one word: the exact value,
    // or a poison value if the from-below root is anything but the exact one or one less (tests/exhaustive_math_sweep.py: all
    // 2^32 inputs)
    {
        const uint32_t q = sqrt_floor_low(x), e = cvt_u32(sqrt_rn_int((float)x));
        out[3 * i + 1] = (q == e || q + 1 == e) ? e : 0xDEAD0001u;
    }
    out[3 * i + 2] = cvt_u32(sqrt_rn_int((float)(int)(x & 0x7FFFFFFFu)) * 10.0f);
}
void launch_math_diag(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_math_diag, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, t.log_thr);
}

// ------------------------------------------------------------------------------------------------
// diagnostics: the fused Mel filterbank term of k_mfcc (sr_dev.h mel_term_fused / mel_tri_of_multiplier) against the
// reference's expression frq_spct[i]*tri[i]/(tri_top/10) in u32 arithmetic (MFCC.C:139-161), for every weight tri in
// [tri_lo, tri_hi) and every energy E in [0, e_max]: blockIdx.y = weight, threads stride over E.  Counts, per weight,
// the E where the two differ (64-bit count; 0 everywhere is the certificate kept in profiles/).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mel_term_sweep(uint32_t tri_lo, uint32_t e_max, unsigned long long *bad)
{
    const uint32_t tri = tri_lo + blockIdx.y, m = mel_fused_multiplier(tri);
    unsigned long long n = 0;
    if (mel_tri_of_multiplier(m) != tri) n += 1ull << 40;  // the literal form would use a wrong weight
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= e_max; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t E = (uint32_t)e;
        const uint32_t ref = E * tri / 100u;  // the reference's u32 expression (no wrap below the bound: checked against 64 bits)
        n += (mel_term_fused(E << 4, m) != ref) || ((uint64_t)E * tri / 100u != ref);
    }
    if (n) atomicAdd(bad + blockIdx.y, n);
}
// diagnostics: the cheap magnitude form (v_sqrt_f32, x10, truncate) against the exact one for n in [0, n_max]
__global__ void __launch_bounds__(256) k_mag_fast_sweep(uint32_t n_max, unsigned long long *bad, uint32_t *first_bad)
{
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= (n_max & 0x7FFFFFFFu); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t n = (uint32_t)i;
        uint32_t want, got;
        if (n_max & 0x80000000u) {  // development: floor of the plain v_sqrt_f32 against the exact (u32)sqrtf (DTW.C:59)
            want = cvt_u32(sqrt_rn_int((float)n));
            got = cvt_u32(__builtin_amdgcn_sqrtf((float)n));
        } else {
            want = cvt_u32(sqrt_rn_int((float)(int)n) * 10.0f);
            got = mag10_small((float)(int)n);
        }
        if (want != got) {
            c++;
            atomicMin(first_bad, n);
        }
    }
    if (c) atomicAdd(bad, c);
}
void launch_mag_fast_sweep(uint32_t n_max, unsigned long long *bad, uint32_t *first_bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_mag_fast_sweep, dim3(1024), dim3(256), 0, s, n_max, bad, first_bad);
}
void launch_mel_term_sweep(uint32_t tri_lo, uint32_t n_tri, uint32_t e_max, unsigned long long *bad, hipStream_t s)
{
    if (!n_tri) return;
    hipLaunchKernelGGL(k_mel_term_sweep, dim3(256, n_tri), dim3(256), 0, s, tri_lo, e_max, bad);
}

// ------------------------------------------------------------------------------------------------
// scalar helpers of DTW.C exposed by the reference-compatible symbols
// ------------------------------------------------------------------------------------------------
__global__ void k_get_dis(const int16_t *pa, const int16_t *pb, uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Frame12 fa = load_frame(pa + (size_t)i * kCoef), fb = load_frame(pb + (size_t)i * kCoef);
    out[i] = get_dis_dev(fa, norm2(fa), fb, norm2(fb));
}
void launch_get_dis(const int16_t *pa, const int16_t *pb, uint32_t *out, uint32_t n, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_get_dis, dim3((n + 63) / 64), dim3(64), 0, s, pa, pb, out, n);
}

__global__ void k_dtw_limit(const uint16_t *xy, uint8_t *out, uint32_t n, int X1, int X2, int in_n, int mdl_n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dtw_out((int)xy[2 * i], (int)xy[2 * i + 1], X1, X2, in_n, mdl_n) ? 1 : 0;
}
void launch_dtw_limit(const uint16_t *xy, uint8_t *out, uint32_t n, int X1, int X2, int in_n, int mdl_n, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_dtw_limit, dim3((n + 63) / 64), dim3(64), 0, s, xy, out, n, X1, X2, in_n, mdl_n);
}

// ------------------------------------------------------------------------------------------------
// diagnostics: fill the WHOLE local data share of every compute unit with a seeded pattern (tests: a kernel that reads LDS it
// has not written -- a table filled behind the last barrier, a region sized one row short -- computes the same values as
// long as the previous tenant of the CU was a workgroup of the same kernel; after this launch it cannot).  One workgroup
// takes all of a CU's LDS (dynamic size = the device's per-workgroup maximum), so at most one is resident per CU; each
// writes its whole allocation and then waits ~20 us so that the dispatcher has to spread the grid over every CU.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lds_poison(uint32_t seed, uint32_t words, uint32_t spin)
{
    extern __shared__ uint32_t poison[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) poison[i] = (seed ^ (i * 2654435761u)) | 0x80000000u;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) acc += poison[i];  // keeps the stores alive
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if (acc == 0x12345678u && seed == 0u) poison[0] = acc;
}
int launch_lds_poison(uint32_t seed, hipStream_t s)
{
    int dev = 0, n_cu = 0, max_lds = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || max_lds <= 0) return -1;
    // all of the CU's LDS for one workgroup (gfx950: 160 KiB); above the default 64 KiB the limit has to be raised explicitly
    if (hipFuncSetAttribute((const void *)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess) {
        (void)hipGetLastError();
        max_lds = 64 * 1024;
    }
    hipLaunchKernelGGL(k_lds_poison, dim3((uint32_t)(4 * n_cu)), dim3(256), (size_t)max_lds, s, seed, (uint32_t)max_lds / 4u, 40000u);
    return max_lds;
}

}  // namespace sr
