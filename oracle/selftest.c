/*
 * TEST INFRASTRUCTURE ONLY -- sanitizer run of the CPU oracle ("oracle under ASan/UBSan").
 *
 *   make -C oracle asan      builds oracle/_ref/oracle_selftest_asan from sr_oracle.c + q15_fft.c + this file with
 *                            -fsanitize=address,undefined -fno-sanitize-recover=all
 *   tests/test_oracle.py::test_oracle_is_clean_under_asan_ubsan runs it.
 *
 * It drives every entry point of sr_oracle.h over synthetic capture buffers (both front ends), extreme feature
 * records (full-scale coefficients: u32 wrap of get_dis, DTW.C:51-57), the edge cases the reference handles by
 * sentinel (silence -> VAD fail, over-long segment -> frm_num 0, gated length ratios -> dis_err, 1-frame records,
 * log(0) frames) and the multi-threaded batch call.  Any out-of-bounds access or undefined operation aborts the
 * process; the checksum printed at the end only keeps the work from being optimised away.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sr_oracle.h"

static uint32_t lcg_state = 12345u;
static uint32_t lcg(void) { return lcg_state = lcg_state * 1664525u + 1013904223u; }
static double urand(void) { return (lcg() >> 8) / 16777216.0; }
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-12)) * cos(6.283185307179586 * urand()); }

/* ADC-like capture: noise head, quiet, a "word" of three drifting sinusoids, quiet */
static void make_capture(uint16_t *buf, uint32_t S, uint32_t fs, uint32_t word_start, uint32_t word_len, double gain)
{
    const double f[3] = {300.0 + 400.0 * urand(), 900.0 + 600.0 * urand(), 2000.0 + 800.0 * urand()};
    for (uint32_t i = 0; i < S; i++) {
        double v = 2048.0 + 6.0 * nrand();
        if (i >= word_start && i < word_start + word_len) {
            const double t = (double)(i - word_start) / fs, env = sin(3.14159265 * (i - word_start) / (double)word_len);
            for (int k = 0; k < 3; k++) v += gain * (400.0 - 100.0 * k) * env * sin(6.283185307 * f[k] * t * (1.0 + 0.1 * t));
            v += 30.0 * nrand();
        }
        buf[i] = (uint16_t)(v < 0 ? 0 : v > 4095 ? 4095 : v);
    }
}

static uint64_t run_front_end(uint32_t fs, uint32_t nfft, uint32_t n_mel, uint32_t max_frames)
{
    sr_oracle_cfg cfg;
    sr_oracle_default_cfg(&cfg);
    cfg.fs = fs;
    cfg.nfft = nfft;
    cfg.n_mel = n_mel;
    cfg.max_frames = max_frames;
    sr_oracle *o = sr_oracle_create(&cfg);
    if (!o) {
        fprintf(stderr, "oracle refused config %u/%u/%u\n", fs, nfft, n_mel);
        exit(2);
    }
    const uint32_t FL = sr_oracle_frame_len(o), HOP = sr_oracle_hop(o), NL = sr_oracle_noise_len(o);
    const uint32_t K = 6, B = 10, S = NL + 8 * FL + HOP * (max_frames + 40);
    uint64_t sum = 0;
    /* templates through the same front end (main.c:121-138) */
    int16_t *tm = calloc((size_t)K * (max_frames + 1) * 12, 2);
    uint32_t tf[6];
    uint8_t tv[6] = {1, 1, 1, 0, 1, 1}; /* one erased slot (main.c:283) */
    uint16_t *cap = malloc((size_t)S * 2 * B);
    for (uint32_t k = 0; k < K; k++) {
        sr_oracle_atap a;
        int32_t seg[6];
        make_capture(cap, S, fs, NL + 5 * FL, HOP * (max_frames / 2 + 7 * k), 1.0);
        sr_oracle_noise_atap(o, cap, NL, &a);
        sr_oracle_vad(o, cap, S, &a, seg);
        tf[k] = (seg[1] >= 0 && seg[0] >= 1) ? sr_oracle_mfcc(o, cap, seg[0], seg[1], &a, tm + (size_t)k * (max_frames + 1) * 12) : 0;
        if (!tf[k]) tv[k] = 0;
        sum += tf[k];
    }
    sr_oracle_templates tpl = {tm, tf, tv, K, (max_frames + 1) * 12};
    /* batch: ordinary words, a silent buffer (VAD fail), a word longer than max_frames (MFCC fail), a loud one */
    for (uint32_t b = 0; b < B; b++) {
        uint32_t len = HOP * (max_frames / 2 + 5 * b);
        double gain = 1.0;
        if (b == 3) len = 0;
        if (b == 4) len = HOP * (max_frames + 20);
        if (b == 5) gain = 6.0; /* clips at the 12-bit rails: u32 wrap of the filterbank products */
        make_capture(cap + (size_t)b * S, S, fs, NL + 5 * FL, len, gain);
    }
    sr_oracle_result res[10];
    int16_t *mf = calloc((size_t)B * max_frames * 12, 2);
    uint32_t sc[10 * 6];
    sr_oracle_recognize_batch(o, cap, S, S, B, &tpl, res, mf, sc, 3);
    for (uint32_t b = 0; b < B; b++) sum += res[b].best_tpl + res[b].min_dis + res[b].status * 7 + res[b].frm_num;
    sr_oracle_result rs[3];
    uint32_t ss[3 * 6];
    sr_oracle_recognize_segments(o, cap, S, &tpl, rs, ss);
    sum += rs[0].min_dis + rs[1].status + rs[2].status;
    /* stage calls with awkward arguments */
    sr_oracle_atap a = {2048, 10, 2, 500};
    sum += (uint64_t)sr_oracle_noise_atap(o, cap, NL - 1, &a); /* silent no-op, VAD.C:33-36 */
    uint32_t mag[512];
    int16_t frame[320] = {0};
    sum += (uint64_t)sr_oracle_fft_mag(o, frame, FL, mag) + mag[0]; /* all-zero frame: log(0) downstream */
    int16_t one[2 * 12];
    for (int i = 0; i < 24; i++) one[i] = (int16_t)(i * 37 - 300);
    sum += sr_oracle_dtw(one, 1, one, 1, 12); /* 1-frame records read the slack row (DTW.C:150-154) */
    free(mf);
    free(cap);
    free(tm);
    sr_oracle_destroy(o);
    return sum;
}

int main(void)
{
    uint64_t sum = 0;
    sum += run_front_end(8000, 1024, 24, 119);
    sum += run_front_end(8000, 1024, 24, 300);
    sum += run_front_end(16000, 512, 40, 64);
    /* feature-level functions on extreme records */
    enum { R = 121 };
    static int16_t a[R * 12], b[R * 12], out[2 * R * 12];
    for (int t = 0; t < 400; t++) {
        const int amp = (t % 3 == 0) ? 32767 : (t % 3 == 1) ? 1500 : 90;
        for (int i = 0; i < R * 12; i++) {
            a[i] = (int16_t)((int)(lcg() % (2u * amp + 1u)) - amp);
            b[i] = (t % 7 == 0) ? a[i] : (int16_t)((int)(lcg() % (2u * amp + 1u)) - amp);
        }
        if (t % 5 == 0)
            for (int i = 0; i < R * 12; i++) a[i] = (lcg() & 1) ? 32767 : -32768, b[i] = (lcg() & 1) ? -32768 : 32767;
        const uint32_t na = 1 + lcg() % 119, nb = 1 + lcg() % 119;
        sum += sr_oracle_get_dis(a, b, 12);
        sum += sr_oracle_dtw(a, na, b, nb, 12);
        sum += sr_oracle_dtw_dp(a, na, b, nb, 12);
        uint32_t frames = 0;
        sum += sr_oracle_get_mdl(a, na, b, nb, 12, out, 2 * R - 2, &frames) + frames;
        sr_oracle_get_mean(a, b, out, 12);
        sum += (uint16_t)out[3];
    }
    uint32_t in[8] = {0u, 1u, 2u, 99u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu}, md[24];
    sr_oracle_math_diag(in, md, 8);
    for (int i = 0; i < 24; i++) sum += md[i];
    printf("oracle selftest ok, checksum %llu\n", (unsigned long long)sum);
    return 0;
}
