/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into the product path.
 *
 * C restatement of the reference's only non-C hot-path component:
 *   Src/BSP/cr4_fft_1024_stm32.s  (ST DSP library, 1024-point radix-4 complex
 *   Q15 FFT for Cortex-M3; entry .s:219-281, butterfly macros .s:95-205,
 *   coefficient table .s:285-629).
 * The image has no ARM toolchain or emulator, so the assembly cannot be run
 * natively; this file follows it instruction group by instruction group.  All
 * register arithmetic is 32-bit two's complement (wraps), LDRSH sign-extends a
 * 16-bit half, STRH keeps the low 16 bits, ASR is an arithmetic (floor) shift.
 *
 * PARITY STATUS of this file: pinned by executing the reference's assembly
 * SOURCE under oracle/arm_fft_interp.py, a small assembler + interpreter for
 * exactly the armasm / Thumb-2 subset the .s file uses (macros expanded, the
 * DCW table laid out, 80 336 instructions per transform).  Where
 * /root/reference exists, tests/test_oracle.py::test_asm_fft_interpreted_
 * equals_restatement requires bit-identical output for every committed golden
 * input (tests/golden/ref_golden.npz: fft_in / fft_out, which is what the GPU
 * tests compare against) and for fresh random, full-scale and zero-padded
 * real inputs.  What remains unexecuted is real Cortex-M3 silicon: the
 * interpreter implements the instruction semantics of the ARMv7-M manual.
 * Further evidence: the regenerated coefficient table equals the .s table
 * entry for entry, and the output agrees with an exact DFT/1024 within the
 * truncation error of the five >>2 passes.
 *
 * The coefficient table is regenerated from its closed form rather than
 * pasted; tests/test_oracle.py::test_twiddles_regenerate_asm_table parses the table out of the .s file
 * (when /root/reference is present) and requires 0 mismatches.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define Q15_NPT 1024

/* packed complex word: real = low half, imag = high half (.s:66-89) */
static inline int32_t lo16(uint32_t w) { return (int16_t)(w & 0xFFFFu); }
static inline int32_t hi16(uint32_t w) { return (int16_t)(w >> 16); }
static inline uint32_t pack16(int32_t re, int32_t im)
{
    return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16);
}
/* wrapping 32-bit helpers (ARM ADD/SUB/MUL/MLA never trap) */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t asr(int32_t a, int n) { return a >> n; }

/*
 * Coefficient table, 3 entries per butterfly, 4 passes (N = 16, 64, 256, 1024).
 * Entry = (Kr', Ki) with Kr' = round(16384*(cos t - sin t)), Ki = round(16384*sin t);
 * per butterfly b the three angles are t = 2*pi*m*b/N for m = 3, 1, 2 in that
 * order (the order the legs j+3q, j+2q, j+q consume them, .s:182-191).
 * 4+16+64+256 = 340 butterflies * 3 = 1020 entries (.s:285-629).
 */
#define Q15_NTW 1020
static int16_t tw_kr[Q15_NTW], tw_ki[Q15_NTW];
static int tw_ready;

static int16_t round_q14(double v)
{
    /* round half away from zero; the table's entries regenerate exactly with it */
    return (int16_t)(v >= 0.0 ? floor(v + 0.5) : -floor(-v + 0.5));
}

void sr_oracle_q15_twiddles(int16_t *kr, int16_t *ki)
{
    static const int mult[3] = { 3, 1, 2 };
    int n = 0;
    for (int N = 16; N <= Q15_NPT; N *= 4) {
        for (int b = 0; b < N / 4; b++) {
            for (int e = 0; e < 3; e++) {
                double t = 2.0 * M_PI * (double)(mult[e] * b) / (double)N;
                kr[n] = round_q14(16384.0 * (cos(t) - sin(t)));
                ki[n] = round_q14(16384.0 * sin(t));
                n++;
            }
        }
    }
}

static void tw_init(void)
{
    if (!tw_ready) {
        sr_oracle_q15_twiddles(tw_kr, tw_ki);
        tw_ready = 1;
    }
}

/* CXMUL_V7 (.s:95-102): YY = Y * conj(K) in Q14 via the 3-multiply trick. */
static inline void cxmul(int32_t *yyr, int32_t *yyi, int32_t yr, int32_t yi, int32_t kr, int32_t ki)
{
    int32_t t2 = wsub(yi, yr);
    int32_t t = wmul(t2, ki);
    t2 = wadd(kr, (int32_t)((uint32_t)ki << 1));
    *yyi = wadd(wmul(yi, kr), t);
    *yyr = wadd(wmul(yr, t2), t);
}

/*
 * Shared radix-4 combine.  s = 0 for the first pass (BUTFLY4ZERO_OPT,
 * .s:147-168), s = 14 for the twiddled passes (CXADDA4 14, .s:105-129).
 * Operates in place on the eight "registers".
 */
static inline void r4_combine(int32_t *ar, int32_t *ai, int32_t *br, int32_t *bi,
                              int32_t *cr, int32_t *ci, int32_t *dr, int32_t *di, int s)
{
    /* (C,D) = (C+D, C-D) */
    *cr = wadd(*cr, *dr);
    *ci = wadd(*ci, *di);
    *dr = wsub(*cr, (int32_t)((uint32_t)*dr << 1));
    *di = wsub(*ci, (int32_t)((uint32_t)*di << 1));
    /* (A,B) = (A+(B>>s), A-(B>>s))/4 */
    *ar = asr(*ar, 2);
    *ai = asr(*ai, 2);
    *ar = wadd(*ar, asr(*br, 2 + s));
    *ai = wadd(*ai, asr(*bi, 2 + s));
    *br = wsub(*ar, asr(*br, 1 + s));
    *bi = wsub(*ai, asr(*bi, 1 + s));
    /* (A,C) = (A+(C>>s)/4, A-(C>>s)/4) */
    *ar = wadd(*ar, asr(*cr, 2 + s));
    *ai = wadd(*ai, asr(*ci, 2 + s));
    *cr = wsub(*ar, asr(*cr, 1 + s));
    *ci = wsub(*ai, asr(*ci, 1 + s));
    /* (B,D) = (B-i*(D>>s)/4, B+i*(D>>s)/4) */
    *br = wadd(*br, asr(*di, 2 + s));
    *bi = wsub(*bi, asr(*dr, 2 + s));
    *di = wsub(*br, asr(*di, 1 + s));
    *dr = wadd(*bi, asr(*dr, 1 + s));
}

static inline unsigned bitrev8(unsigned v)
{
    v = ((v & 0xF0u) >> 4) | ((v & 0x0Fu) << 4);
    v = ((v & 0xCCu) >> 2) | ((v & 0x33u) << 2);
    v = ((v & 0xAAu) >> 1) | ((v & 0x55u) << 1);
    return v;
}

/* Same symbol, same arguments as the assembly routine (MFCC.C:12, .s:22). */
void cr4_fft_1024_stm32(void *pssOUT, void *pssIN, uint16_t Nbin)
{
    uint32_t *out = (uint32_t *)pssOUT;
    const uint32_t *in = (const uint32_t *)pssIN;
    (void)Nbin; /* the routine only converts 1024 points (.s:214-215) */
    tw_init();

    /* pass 1, preloop_v7 (.s:226-232): bit-reversed gather, no twiddles.
       RBIT(index)>>22 is a byte offset = 4*bitrev8(index); the four legs are
       loaded NPT bytes (256 words) apart in the order A, C, B, D (.s:134-145). */
    for (unsigned idx = 0; idx < 256; idx++) {
        unsigned r = bitrev8(idx);
        int32_t ar = lo16(in[r]), ai = hi16(in[r]);
        int32_t cr = lo16(in[r + 256]), ci = hi16(in[r + 256]);
        int32_t br = lo16(in[r + 512]), bi = hi16(in[r + 512]);
        int32_t dr = lo16(in[r + 768]), di = hi16(in[r + 768]);
        r4_combine(&ar, &ai, &br, &bi, &cr, &ci, &dr, &di, 0);
        out[4 * idx + 0] = pack16(ar, ai);
        out[4 * idx + 1] = pack16(br, bi);
        out[4 * idx + 2] = pack16(cr, ci);
        out[4 * idx + 3] = pack16(di, dr); /* "inversion here" (.s:176-177) */
    }

    /* passes 2..5, passloop_v7 (.s:254-279), in place on the output array.
       q = quarter-group length in words (index/4 in the asm); the coefficient
       pointer rewinds after every group but the last of a pass (.s:268-274). */
    int tw_base = 0;
    for (int q = 4; q < Q15_NPT; q *= 4) {
        for (int g = 0; g < Q15_NPT; g += 4 * q) {
            int k = tw_base;
            for (int b = 0; b < q; b++, k += 3) {
                int j = g + b;
                int32_t ar, ai, br, bi, cr, ci, dr, di;
                uint32_t w;
                w = out[j + 3 * q];
                cxmul(&dr, &di, lo16(w), hi16(w), tw_kr[k + 0], tw_ki[k + 0]);
                w = out[j + 2 * q];
                cxmul(&cr, &ci, lo16(w), hi16(w), tw_kr[k + 1], tw_ki[k + 1]);
                w = out[j + q];
                cxmul(&br, &bi, lo16(w), hi16(w), tw_kr[k + 2], tw_ki[k + 2]);
                w = out[j];
                ar = lo16(w);
                ai = hi16(w);
                r4_combine(&ar, &ai, &br, &bi, &cr, &ci, &dr, &di, 14);
                out[j] = pack16(ar, ai);
                out[j + q] = pack16(br, bi);
                out[j + 2 * q] = pack16(cr, ci);
                out[j + 3 * q] = pack16(di, dr); /* inversion (.s:203-204) */
            }
        }
        tw_base += 3 * q;
    }
}

/* ------------------------------------------------------------------------------------------------
 * EXTENSION -- no reference counterpart (the reference's FFT "can only convert 1024 points", .s:214-215;
 * BASELINE.json configs[4] asks for a 512-point front end).  Definition used by oracle and device alike:
 *
 *   fft256: the same ST radix-4 structure at 256 points: bit-reversed first pass (legs 64 words apart, loaded
 *           A, C, B, D as in .s:134-145) + three twiddled passes with the N = 16, 64, 256 blocks of the same
 *           coefficient table; output scaled 1/256 with the same per-pass truncation.
 *   fft512: E = fft256(even samples), O = fft256(odd samples), then one truncating radix-2 pass
 *           X[k]     = (E[k] + O[k]*conj(W[k])) >> 1      W[k] = Q14 (cos, sin)(2*pi*k/512), rounded half away,
 *           X[k+256] = (E[k] - O[k]*conj(W[k])) >> 1      product O*conj(W) taken >> 14 (arithmetic) before the add;
 *           results stored as 16+16 bits like every other pass.  Output scaled 1/512.
 * ------------------------------------------------------------------------------------------------ */
static inline unsigned bitrev6(unsigned v) { return bitrev8(v) >> 2; }

void sr_oracle_q15_fft256(uint32_t *out, const uint32_t *in)
{
    tw_init();
    for (unsigned idx = 0; idx < 64; idx++) {
        unsigned r = bitrev6(idx);
        int32_t ar = lo16(in[r]), ai = hi16(in[r]);
        int32_t cr = lo16(in[r + 64]), ci = hi16(in[r + 64]);
        int32_t br = lo16(in[r + 128]), bi = hi16(in[r + 128]);
        int32_t dr = lo16(in[r + 192]), di = hi16(in[r + 192]);
        r4_combine(&ar, &ai, &br, &bi, &cr, &ci, &dr, &di, 0);
        out[4 * idx + 0] = pack16(ar, ai);
        out[4 * idx + 1] = pack16(br, bi);
        out[4 * idx + 2] = pack16(cr, ci);
        out[4 * idx + 3] = pack16(di, dr);
    }
    int tw_base = 0;
    for (int q = 4; q < 256; q *= 4) {
        for (int g = 0; g < 256; g += 4 * q) {
            int k = tw_base;
            for (int b = 0; b < q; b++, k += 3) {
                int j = g + b;
                int32_t ar, ai, br, bi, cr, ci, dr, di;
                uint32_t w;
                w = out[j + 3 * q];
                cxmul(&dr, &di, lo16(w), hi16(w), tw_kr[k + 0], tw_ki[k + 0]);
                w = out[j + 2 * q];
                cxmul(&cr, &ci, lo16(w), hi16(w), tw_kr[k + 1], tw_ki[k + 1]);
                w = out[j + q];
                cxmul(&br, &bi, lo16(w), hi16(w), tw_kr[k + 2], tw_ki[k + 2]);
                w = out[j];
                ar = lo16(w);
                ai = hi16(w);
                r4_combine(&ar, &ai, &br, &bi, &cr, &ci, &dr, &di, 14);
                out[j] = pack16(ar, ai);
                out[j + q] = pack16(br, bi);
                out[j + 2 * q] = pack16(cr, ci);
                out[j + 3 * q] = pack16(di, dr);
            }
        }
        tw_base += 3 * q;
    }
}

void sr_oracle_q15_w512(int16_t *wc, int16_t *ws)
{
    for (int k = 0; k < 256; k++) {
        double t = 2.0 * M_PI * (double)k / 512.0;
        wc[k] = round_q14(16384.0 * cos(t));
        ws[k] = round_q14(16384.0 * sin(t));
    }
}

void sr_oracle_q15_fft512(uint32_t *out, const uint32_t *in)
{
    static int16_t wc[256], ws[256];
    static int w_ready;
    uint32_t ev[256], od[256], E[256], O[256];
    if (!w_ready) {
        sr_oracle_q15_w512(wc, ws);
        w_ready = 1;
    }
    for (int i = 0; i < 256; i++) {
        ev[i] = in[2 * i];
        od[i] = in[2 * i + 1];
    }
    sr_oracle_q15_fft256(E, ev);
    sr_oracle_q15_fft256(O, od);
    for (int k = 0; k < 256; k++) {
        int32_t orr = lo16(O[k]), oi = hi16(O[k]);
        /* O * conj(W) in Q14, then >> 14 */
        int32_t pr = asr(wadd(wmul(orr, wc[k]), wmul(oi, ws[k])), 14);
        int32_t pi = asr(wsub(wmul(oi, wc[k]), wmul(orr, ws[k])), 14);
        int32_t er = lo16(E[k]), ei = hi16(E[k]);
        out[k] = pack16(asr(wadd(er, pr), 1), asr(wadd(ei, pi), 1));
        out[k + 256] = pack16(asr(wsub(er, pr), 1), asr(wsub(ei, pi), 1));
    }
}
