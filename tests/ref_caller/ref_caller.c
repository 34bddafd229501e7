/*
 * TEST INFRASTRUCTURE -- the "swap the objects" proof of INTEGRATION.md section 1.
 *
 * A caller written against the REFERENCE's OWN headers -- Src/Speech_Recog/VAD.H, MFCC.H and DTW.H, found on the
 * include path where they lie under /root/reference (plus Src/BSP/ADC.H for fs / VcBuf_Len / atap_len, and the
 * six-typedef oracle/shim/stm32f10x.h standing in for the vendor header) -- and NOT against include/sr_engine.h.
 * It is linked with -lsr_engine instead of the reference's VAD.o / MFCC.o / DTW.o / cr4_fft_1024_stm32.o: if the
 * prototypes, struct layouts or constants of the library differed from the reference's, this file would not
 * compile, not link, or not print the golden results.
 *
 * What it does is the firmware's recognition sequence (Src/APP/main.c:258-295): adapt to the noise head, find the
 * spoken segment, extract the MFCC record of segment 0, score it against every valid slot of the template store
 * and keep the first smallest distance.  The store is an array in memory (a file image of the flash window of
 * Src/BSP/Flash.H:11-20: 4 KiB per slot, valid slots marked 12345) instead of an absolute flash address.
 *
 *   ref_caller store.bin capture.bin [capture.bin ...]      captures: VcBuf_Len u16 samples each
 *
 * Built by oracle/Makefile (target _ref/ref_caller) where the reference tree exists; the binary travels to the GPU
 * box with the snapshot like oracle/_ref/libsr_ref.so; tests/test_gpu_parity.py runs it on the golden captures.
 */
#include "stm32f10x.h"
#include "ADC.h"
#include "VAD.H"
#include "MFCC.H"
#include "DTW.H"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SLOT_BYTES 4096u /* size_per_ftr, Flash.H:14 */
#define SLOT_VALID 12345u /* save_mask, Flash.H:11 */

/* the layouts this translation unit was compiled with are the reference's (MFCC.H:18-25, VAD.H:10-22) */
typedef char ftr_is_2860_bytes[(sizeof(v_ftr_tag) == 4 + 2 * vv_frm_max * mfcc_num && vv_frm_max == 119) ? 1 : -1];
typedef char atap_is_12_bytes[(sizeof(atap_tag) == 12) ? 1 : -1];

static void *slurp(const char *path, size_t *bytes)
{
    FILE *f = fopen(path, "rb");
    void *p;
    long n;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    n = ftell(f);
    fseek(f, 0, SEEK_SET);
    p = malloc((size_t)n + 2);
    if (p && fread(p, 1, (size_t)n, f) != (size_t)n) {
        free(p);
        p = NULL;
    }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char **argv)
{
    size_t store_bytes = 0, cap_bytes = 0;
    u8 *store;
    u32 n_slots;
    int a;

    if (argc < 3) {
        fprintf(stderr, "usage: %s store.bin capture.bin [...]\n", argv[0]);
        return 2;
    }
    store = (u8 *)slurp(argv[1], &store_bytes);
    if (!store || store_bytes < SLOT_BYTES) {
        fprintf(stderr, "cannot read the template store %s\n", argv[1]);
        return 2;
    }
    n_slots = (u32)(store_bytes / SLOT_BYTES);
    printf("reference headers: sizeof(v_ftr_tag)=%u vv_frm_max=%u VcBuf_Len=%u atap_len=%u; store: %u slots\n",
           (unsigned)sizeof(v_ftr_tag), (unsigned)vv_frm_max, (unsigned)VcBuf_Len, (unsigned)atap_len, (unsigned)n_slots);

    for (a = 2; a < argc; a++) {
        /* one sample of head room in front: get_mfcc reads the sample before the segment (MFCC.C:119) */
        u16 *raw = (u16 *)slurp(argv[a], &cap_bytes), *buf;
        atap_tag atap;
        valid_tag voice[max_vc_con];
        v_ftr_tag ftr;
        u32 best = 0, best_dis = dis_max, slot;

        if (!raw || cap_bytes < 2u * VcBuf_Len) {
            fprintf(stderr, "capture %s: need %u samples\n", argv[a], (unsigned)VcBuf_Len);
            return 2;
        }
        buf = raw;
        memset(&ftr, 0, sizeof ftr);
        noise_atap(buf, atap_len, &atap);
        VAD(buf, VcBuf_Len, voice, &atap);
        if (voice[0].end == NULL) {
            printf("%s: VAD fail slot=-1 dis=%u\n", argv[a], (unsigned)dis_err);
            free(raw);
            continue;
        }
        get_mfcc(&voice[0], &ftr, &atap);
        if (ftr.frm_num == 0) {
            printf("%s: MFCC fail slot=-1 dis=%u\n", argv[a], (unsigned)dis_err);
            free(raw);
            continue;
        }
        for (slot = 0; slot < n_slots; slot++) {
            v_ftr_tag *mdl = (v_ftr_tag *)(store + (size_t)slot * SLOT_BYTES);
            u32 d = (mdl->save_sign == SLOT_VALID) ? dtw(&ftr, mdl) : dis_err;
            if (d < best_dis) {
                best_dis = d;
                best = slot;
            }
        }
        printf("%s: slot=%u dis=%u frm_num=%u seg=[%ld,%ld) mid=%u n_thl=%u z_thl=%u s_thl=%u mfcc0=%d,%d,%d\n", argv[a],
               (unsigned)best, (unsigned)best_dis, (unsigned)ftr.frm_num, (long)(voice[0].start - buf), (long)(voice[0].end - buf),
               (unsigned)atap.mid_val, (unsigned)atap.n_thl, (unsigned)atap.z_thl, (unsigned)atap.s_thl, ftr.mfcc_dat[0],
               ftr.mfcc_dat[1], ftr.mfcc_dat[2]);
        free(raw);
    }
    free(store);
    return 0;
}
