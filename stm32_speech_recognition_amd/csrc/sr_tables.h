// Constant tables of the recognition front end, generated on the host at sr_create() and
// uploaded once to HBM.  Reference: Src/Speech_Recog/MFCC_Arg.h:6-44 (pasted Matlab output),
// generator formulas Matlab/matlab仿真/speech_recog.m:217-310 and teat.m:19-27; FFT coefficient
// table Src/BSP/cr4_fft_1024_stm32.s:285-629.
#pragma once
#include <cstdint>
#include <vector>

namespace sr {

constexpr int kFrameLen = 160;  // VAD.H:7 at fs = 8000
constexpr int kHop = 80;        // frame_len - frame_mov, VAD.H:8
constexpr int kNfft = 1024;     // MFCC.H:8
constexpr int kBins = 512;      // frq_max, MFCC.H:9
constexpr int kMel = 24;        // tri_num, MFCC.H:12
constexpr int kCoef = 12;       // mfcc_num, MFCC.H:13
constexpr int kTwiddles = 1020; // 3 per butterfly, passes N = 16, 64, 256, 1024
constexpr int kLogMax = 2218;   // floor(100*ln(2^32-1))
constexpr int kTieMax = 32768;  // entries of the DTW tie-threshold table (roots below this take the one-root step)

// Mel filterbank term E * tri / (tri_top / 10) of MFCC.C:139-161 with the division folded into the multiplier (round 5):
//     floor(E * tri / 100) == mul_hi(E << 4, M)   with   M = ceil(tri * 2^28 / 100)   for every E <= kMelFusedMaxE.
// E*M / 2^28 = E*tri/100 + E*d/2^28 with 0 <= d < 1; the fractional part of E*tri/100 is at most 99/100, so the quotient
// is unchanged while E*d/2^28 < 1/100, i.e. E < 2^28/100.  tri <= tri_top = 1000 keeps M below 2^32 (it needs
// tri < 1600) and E*tri below 2^32 (no u32 wrap of the reference's product to mirror); E << 4 needs E < 2^28.
// Certified by exhaustion on the device for every tri in 0..1000 and every E in 0..kMelFusedMaxE (sr_mel_term_sweep,
// profiles/r05_mel_term_sweep.txt).  A frame with a larger E anywhere takes the literal form, with the weight
// recovered as tri = mul_hi(M, 1600) = floor(M * 100 / 2^28) (exact: M*100/2^28 - tri = 100*d/2^28 < 1).

// Windowing, (s16)(temp * hamm[i] / (hamm_top / 10)) of MFCC.C:122, with the division folded into the multiplier (round 5):
//     trunc(t * h / 1000) == (t << 5) * M + (t < 0 ? 2^32 - 1 : 0)  >> 32     with   M = ceil(h * 2^27 / 1000)
// (64-bit signed arithmetic, arithmetic shift: ONE v_mad_i64_i32 whose 64-bit addend is the sign mask of t, zero-extended).
// (t << 5) * M / 2^32 = t*h/1000 + t*d/2^27 with 0 <= d < 1, and |t| <= 65535 + 62258 = 127793 (a u16 sample minus a u16
// mid value, minus 95/100 of another one) keeps |t*d/2^27| below 0.000953 < 1/1000: for t >= 0 the floor of the shift is the
// quotient; for t < 0 adding 2^32 - 1 before the shift turns the floor into a ceiling, which is C's truncation toward zero.
// h <= hamm_top = 10000 keeps M below 2^31.  Checked for every t of the domain x every window weight of the three front ends
// on the host (tests/test_oracle.py::test_window_fused_multiplier_is_exact) and through the golden MFCC fixtures.
constexpr int kWinShift = 5;
constexpr int32_t hamm_fused_multiplier(uint32_t h) { return (int32_t)((((uint64_t)h << 27) + 999u) / 1000u); }

constexpr uint32_t kMelFusedMaxE = (1u << 28) / 100u;  // 2 684 354  (|X|*10 <= 1638)
// re^2 + im^2 of a bin up to which (a) |X|*10 = (u32)(sqrtf(n)*10) <= 1638, hence E <= 1638^2 = 2 683 044 <= kMelFusedMaxE, and
// (b) the uncorrected v_sqrt_f32 gives the same (u32)(sqrtf(n)*10) as the exact root (true for every n <= 70 171 on gfx950:
// sr_mag_fast_sweep, repeated by every -m gpu run): 10*sqrt(26843) = 1638.38.  One test on n serves both fast forms of k_mfcc.
constexpr uint32_t kMagSmallMax = 26843;
static_assert(1638u * 1638u <= kMelFusedMaxE && 100ull * (kMagSmallMax + 1) < 1639ull * 1639ull, "quiet-frame bound");
// re^2 + im^2 of a bin up to which the uncorrected v_sqrt_f32 gives the same (u32)(sqrtf(n)*10) as the exact root ON gfx950 (the
// first difference is at n = 70 172): k_mfcc's MID tier (round 6: cheap magnitude, literal filterbank term).  A property of the
// chip's v_sqrt_f32, not of the arithmetic: sr_create sweeps [0, kMagCheapMax] on the device it runs on (k_mag_fast_sweep) and
// hands the kernel a bound of 0 -- every frame then takes the exactly corrected root -- if the sweep finds a difference.
constexpr uint32_t kMagCheapMax = 70171;
static_assert(kMagSmallMax <= kMagCheapMax, "the quiet tier lies inside the cheap-magnitude range");
constexpr uint32_t kMelTriMax = 1599;                  // largest weight whose multiplier fits 32 bits
constexpr uint32_t mel_fused_multiplier(uint32_t tri) { return (uint32_t)((((uint64_t)tri << 28) + 99u) / 100u); }

// The two front ends the kernels are built for: the reference's (ADC.H:7, VAD.H:5-8, MFCC.H:7-13) and the
// 16 kHz / 512-point / 40-Mel EXTENSION of BASELINE.json configs[4] (no reference counterpart).
struct FrontEnd {
    int fs, frame_len, hop, nfft, bins, n_mel, n_coef;
    bool generic;  // neither of the two specialised kernels: k_mfcc_gen + the VAD instance of the framing
};
constexpr FrontEnd kFrontRef = {8000, 160, 80, 1024, 512, 24, 12, false};
constexpr FrontEnd kFrontExt = {16000, 320, 160, 512, 256, 40, 12, false};

struct HostTables {
    std::vector<uint16_t> hamm;      // [160]
    std::vector<uint16_t> tri_cen;   // [24]
    std::vector<uint16_t> tri_even;  // [512]
    std::vector<uint16_t> tri_odd;   // [512]
    std::vector<int8_t> dct;         // [12*24]
    // Per twiddle K = Kc + i*Ks (Q14; Kc = Kr' + Ki of the ST table) two packed words so that
    // Y*conj(K) is two 16-bit dot products of the packed sample (re lo, im hi):
    //   tw_a = (Kc, Ks)   -> re' = Yr*Kc + Yi*Ks
    //   tw_b = (-Ks, Kc)  -> im' = Yi*Kc - Yr*Ks
    std::vector<uint32_t> tw_a, tw_b;  // [1020]
    std::vector<int16_t> tw_kr, tw_ki; // raw (Kr', Ki) columns as in the .s table (exported by sr_build_tables, diffed by tests)
    // log_thr[m] = smallest n with (u32)(log((double)n)*100) >= m, m = 0..2218; [2219] = sentinel.
    std::vector<uint32_t> log_thr;
    // EXTENSION only: Q14 (cos, sin)(2*pi*k/512), k < 256, packed like the butterfly coefficients
    std::vector<uint32_t> w512_a, w512_b;
    // DTW step (DTW.C:156-184): with g(d) = (u32)sqrtf((float)d) (DTW.C:59), T(g) = min{d : g(d) >= g + 1} is where the
    // root steps from g to g + 1.  tie_delta[g] = T(g) - (g + 1)^2 + 1 for g < kTieMax, so that
    // T(g) = g*(g + 2) + tie_delta[g] (one 24-bit multiply-add on the device).  Exact squares up to 2^24 convert to
    // float exactly, so the entry is 1 for g < 4096; above, (float)d rounds d and T(g) falls a little short of (g + 1)^2.
    std::vector<int8_t> tie_delta;
};

void build_tables(HostTables &t, const FrontEnd &fe);
// entries of the log step table where this host's libm disagreed with the shipped positions at the last build_tables
int log_table_mismatches();

}  // namespace sr
