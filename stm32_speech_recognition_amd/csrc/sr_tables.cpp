#include "sr_device.h"
#include "sr_tables.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace sr {

// Step positions of (u32)(log((double)n)*100) as the libm of the image the golden fixtures were generated in evaluates
// them (gen_log_thr_ref.py).  The table built from the run-time host's libm is compared with it, see gen_log_thr.
static const uint32_t kLogThrRef[kLogMax + 2] = {
#include "sr_log_thr_ref.inc"
};
static std::atomic<int> g_log_mismatches{0};
int log_table_mismatches() { return g_log_mismatches.load(); }

// Matlab int32(): round half away from zero.
static double mround(double v) { return v >= 0.0 ? std::floor(v + 0.5) : -std::floor(-v + 0.5); }

// Hamming window scaled to hamm_top = 10000 (speech_recog.m:217-222, MFCC_Arg.h:6-9).
static void gen_hamm(HostTables &t, const FrontEnd &fe)
{
    t.hamm.resize(fe.frame_len);
    for (int i = 0; i < fe.frame_len; i++)
        t.hamm[i] = (uint16_t)mround((0.54 - 0.46 * std::cos(2 * M_PI * i / (fe.frame_len - 1))) * 10000.0);
}

// Mel triangle centres and the two interleaved triangle poly-lines (speech_recog.m:240-310).
// The Matlab arrays are 1-based; element j is C index j-1, and Matlab's "odd" poly-line is the
// C table tri_even (filters 0,2,4,...) and vice versa (MFCC_Arg.h:17-27).
static void gen_tri(HostTables &t, const FrontEnd &fe)
{
    const double top = 1000.0, fs = fe.fs;
    const int n = fe.n_mel, nb = fe.bins;
    const double f_max = fs / 2;
    const double mel_max = 2595 * std::log10(1 + f_max / 700);
    const double mel_step = mel_max / (n + 1);
    std::vector<double> cen(n + 2, 0.0), lineA(nb + 2, 0.0), lineB(nb + 2, 0.0);
    t.tri_cen.resize(n);
    for (int i = 1; i <= n; i++) {
        double c = (i < 1000.0 / mel_step) ? mel_step * i : (std::exp(std::log(10) * (mel_step * i) / 2595) - 1) * 700;
        cen[i] = mround(c / (f_max / nb));
        t.tri_cen[i - 1] = (uint16_t)cen[i];
    }
    auto rise = [&](std::vector<double> &a, int lo, int hi) {
        for (long j = (long)cen[lo]; j <= (long)cen[hi]; j++)
            if (j >= 1 && j <= nb) a[j] = top * (j - cen[lo]) / (cen[hi] - cen[lo]);
    };
    auto fall = [&](std::vector<double> &a, int lo, int hi) {
        for (long j = (long)cen[lo] + 1; j <= (long)cen[hi]; j++)
            if (j >= 1 && j <= nb) a[j] = top * (cen[hi] - j) / (cen[hi] - cen[lo]);
    };
    for (long j = 1; j <= (long)cen[1]; j++) lineA[j] = top * j / cen[1];
    fall(lineA, 1, 2);
    for (int h = 3; h <= n; h += 2) {
        rise(lineA, h - 1, h);
        fall(lineA, h, h + 1);
    }
    for (int h = 2; h + 2 <= n; h += 2) {
        rise(lineB, h - 1, h);
        fall(lineB, h, h + 1);
    }
    rise(lineB, n - 1, n);
    for (long j = (long)cen[n] + 1; j <= nb; j++) lineB[j] = top * (nb - j) / (nb - cen[n]);
    t.tri_even.resize(nb);
    t.tri_odd.resize(nb);
    for (int j = 1; j <= nb; j++) {
        t.tri_even[j - 1] = (uint16_t)mround(lineA[j]);
        t.tri_odd[j - 1] = (uint16_t)mround(lineB[j]);
    }
}

// 12 x 24 cosine table scaled by 100 (teat.m:19-26, MFCC_Arg.h:29-44).
static void gen_dct(HostTables &t, const FrontEnd &fe)
{
    const int nm = fe.n_mel;
    t.dct.resize(fe.n_coef * nm);
    for (int h = 1; h <= fe.n_coef; h++)
        for (int j = 1; j <= nm; j++)
            t.dct[(h - 1) * nm + (j - 1)] = (int8_t)mround(std::cos(h * M_PI * (j - 0.5) / nm) * 100);
}

// EXTENSION: coefficients of the final radix-2 pass of the 512-point transform (see oracle/q15_fft.c)
static void gen_w512(HostTables &t)
{
    t.w512_a.resize(256);
    t.w512_b.resize(256);
    for (int k = 0; k < 256; k++) {
        const double th = 2.0 * M_PI * k / 512.0;
        const int wc = (int)mround(16384.0 * std::cos(th)), ws = (int)mround(16384.0 * std::sin(th));
        t.w512_a[k] = ((uint32_t)wc & 0xFFFFu) | ((uint32_t)ws << 16);
        t.w512_b[k] = ((uint32_t)(-ws) & 0xFFFFu) | ((uint32_t)wc << 16);
    }
}

// Coefficients of the ST radix-4 FFT: per pass N in {16,64,256,1024}, per butterfly b < N/4, three
// entries for the legs j+3q, j+2q, j+q with angles 2*pi*m*b/N, m = 3, 1, 2
// (cr4_fft_1024_stm32.s:182-191 consumption order; table .s:285-629).  The table stores
// Kr' = round(16384(cos-sin)), Ki = round(16384 sin); the butterfly needs Kc = Kr' + Ki and Ks = Ki.
static void gen_twiddles(HostTables &t)
{
    static const int mult[3] = {3, 1, 2};
    t.tw_a.resize(kTwiddles);
    t.tw_b.resize(kTwiddles);
    t.tw_kr.resize(kTwiddles);
    t.tw_ki.resize(kTwiddles);
    int n = 0;
    for (int N = 16; N <= kNfft; N *= 4)
        for (int b = 0; b < N / 4; b++)
            for (int e = 0; e < 3; e++, n++) {
                double th = 2.0 * M_PI * (mult[e] * b) / N;
                int kr = (int)mround(16384.0 * (std::cos(th) - std::sin(th)));
                int ki = (int)mround(16384.0 * std::sin(th));
                int kc = kr + ki, ks = ki;
                t.tw_kr[n] = (int16_t)kr;
                t.tw_ki[n] = (int16_t)ki;
                t.tw_a[n] = ((uint32_t)kc & 0xFFFFu) | ((uint32_t)ks << 16);
                t.tw_b[n] = ((uint32_t)(-ks) & 0xFFFFu) | ((uint32_t)kc << 16);
            }
}

// MFCC.C:168 takes (u32)(log((double)n)*100) of a u32 filterbank output.  The device evaluates it
// as a step function: an fp32 estimate of the step index corrected against exact step positions.
// The positions are found here with the host's own libm double log, i.e. the very expression the
// reference C path evaluates on this machine, so the device result equals it for every u32.
static uint32_t log100(uint32_t n) { return n ? (uint32_t)(std::log((double)n) * 100) : 0u; }

static void gen_log_thr(HostTables &t)
{
    t.log_thr.assign(kLogMax + 2, 0xFFFFFFFFu);
    t.log_thr[0] = 1;
    for (int m = 1; m <= kLogMax; m++) {
        uint64_t lo = t.log_thr[m - 1], hi = 0xFFFFFFFFull;  // first n in (lo, hi] with log100(n) >= m
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (log100((uint32_t)mid) >= (uint32_t)m)
                hi = mid;
            else
                lo = mid + 1;
        }
        t.log_thr[m] = (uint32_t)lo;
    }
    // Exactness independent of the deployment host: `log` is not correctly rounded by every libm, and a last-bit
    // difference at one of the 2219 integer crossings of log(n)*100 would move a step by one n.  The positions found with
    // THIS host's libm are compared with the shipped ones (the libm the golden fixtures were made with): on a mismatch the
    // shipped table is used -- results then equal the fixtures', not this host's C path -- and the count is kept for
    // sr_log_table_mismatches() / the warning of sr_create.  Development hooks (sr_dev_hook): "log_thr_from_host" keeps the
    // host's table (used to regenerate the .inc); "perturb_log_thr" = m (tests) moves the host-built entry m by one to
    // exercise the check.
    {
        const int64_t m = dev_hook(kHookPerturbLogThr);
        if (m >= 1 && m <= kLogMax) t.log_thr[m] += 1;
    }
    const bool keep = dev_hook(kHookLogThrFromHost) != 0;
    int bad = 0;
    for (int m = 0; m <= kLogMax + 1; m++) bad += t.log_thr[m] != kLogThrRef[m];
    g_log_mismatches.store(bad);
    if (bad && !keep) t.log_thr.assign(kLogThrRef, kLogThrRef + kLogMax + 2);
}

// DTW.C:59 takes (u32)sqrtf((float)d) of a u32 sum of squares.  The root function g is monotone; the staged DTW kernel
// decides the reference's tie order (DTW.C:168-184) on the squared distances against T(g) = first d whose root is g + 1,
// found here with the host's own float conversion and sqrtf -- the very expression the reference evaluates (both are
// IEEE-exact operations, so the table does not depend on the libm).
static uint32_t root_u32(uint32_t d) { return (uint32_t)sqrtf((float)d); }

static void gen_tie_delta(HostTables &t)
{
    t.tie_delta.assign(kTieMax, 1);
    for (uint32_t g = 4095; g < (uint32_t)kTieMax; g++) {
        const uint64_t M = (uint64_t)(g + 1) * (g + 1);  // <= 2^30
        // g(M) >= g + 1 always ((float)M is within half a float spacing of M, far less than the 2g + 1 to the next
        // square); walk down to the first d that still has the root g + 1.  (float)d is constant over runs of up to 64
        // integers here, T(g) > g^2 + g, so the walk is short.
        uint32_t d = (uint32_t)M;
        while (root_u32(d - 1) >= g + 1) d--;
        const int64_t delta = (int64_t)d - (int64_t)M + 1;
        t.tie_delta[g] = (int8_t)delta;
        if (delta < -128 || delta > 127) t.tie_delta.clear();  // cannot happen below 2^30 (half a float spacing <= 32)
    }
}

void build_tables(HostTables &t, const FrontEnd &fe)
{
    gen_tie_delta(t);
    gen_hamm(t, fe);
    gen_tri(t, fe);
    gen_dct(t, fe);
    gen_twiddles(t);
    gen_log_thr(t);
    gen_w512(t);
}

}  // namespace sr
