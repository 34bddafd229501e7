/*
 * Reference-side call pattern against libsr_engine.so -- plain C, no Python, no torch.
 *
 * The body of recognise() is main.c:258-283 of the reference VERBATIM in structure (noise_atap -> VAD ->
 * get_mfcc -> dtw over the slots, strict '<' argmin); the only new lines are the two hand-overs the firmware
 * gets from absolute addresses (template store, Flash.H:19-20).  The same capture is then recognised through
 * the one-call drop-in spch_recg() and through the batched API, and the three answers must agree.
 *
 *   gcc -std=gnu99 -Iinclude examples/spch_recg_demo.c -Lstm32_speech_recognition_amd -lsr_engine \
 *       -Wl,-rpath,$PWD/stm32_speech_recognition_amd -o /tmp/spch_recg_demo
 *   /tmp/spch_recg_demo store.bin capture.bin        (store: n x 4096 bytes, capture: 16000 x u16)
 */
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "sr_engine.h"

#define VcBuf_Len 16000 /* ADC.H:9  */
#define atap_len 2400   /* ADC.H:11 */
#define size_per_ftr 4096
#define save_mask 12345
#define dis_err 0xFFFFFFFFu

static uint16_t VcBuf[VcBuf_Len];
static atap_tag atap_arg;
static valid_tag valid_voice[3];
static v_ftr_tag ftr;

/* main.c:249-296 with the flash window replaced by a pointer */
static int recognise(uint16_t *v_dat, const uint8_t *store, unsigned n_slots, uint32_t *mtch_dis)
{
    unsigned i, min_comm = 0;
    uint32_t min_dis = dis_err, cur_dis;
    noise_atap(v_dat, atap_len, &atap_arg);
    VAD(v_dat, VcBuf_Len, valid_voice, &atap_arg);
    if (valid_voice[0].end == (void *)0) {
        *mtch_dis = dis_err;
        return -1;
    }
    get_mfcc(&(valid_voice[0]), &ftr, &atap_arg);
    if (ftr.frm_num == 0) {
        *mtch_dis = dis_err;
        return -1;
    }
    for (i = 0; i < n_slots; i++) {
        v_ftr_tag *ftr_mdl = (v_ftr_tag *)(store + (size_t)i * size_per_ftr);
        cur_dis = ((ftr_mdl->save_sign) == save_mask) ? dtw(&ftr, ftr_mdl) : dis_err;
        if (cur_dis < min_dis) {
            min_dis = cur_dis;
            min_comm = i;
        }
    }
    *mtch_dis = min_dis;
    return (int)min_comm;
}

int main(int argc, char **argv)
{
    FILE *f;
    long sz;
    uint8_t *store;
    unsigned n_slots;
    uint32_t d1 = 0, d2 = 0;
    int slot;
    uint8_t *label;
    sr_result res;

    if (argc < 3) {
        fprintf(stderr, "usage: %s store.bin capture.bin\n", argv[0]);
        return 2;
    }
    f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    store = malloc((size_t)sz);
    if (fread(store, 1, (size_t)sz, f) != (size_t)sz) return 2;
    fclose(f);
    n_slots = (unsigned)(sz / size_per_ftr);
    f = fopen(argv[2], "rb");
    if (!f || fread(VcBuf, 2, VcBuf_Len, f) != VcBuf_Len) return 2;
    fclose(f);

    /* (1) the reference's own call sequence, function by function */
    slot = recognise(VcBuf, store, n_slots, &d1);

    /* (2) the one-call drop-in of main.c:336 */
    if (sr_compat_set_templates(store, n_slots, size_per_ftr) != SR_OK) {
        fprintf(stderr, "%s\n", sr_last_error());
        return 1;
    }
    label = spch_recg(VcBuf, &d2);

    /* (3) the batched API on the same engine */
    if (sr_recognize_batch(sr_compat_engine(), VcBuf, VcBuf_Len, VcBuf_Len, 1, &res, NULL, NULL, NULL) != SR_OK) {
        fprintf(stderr, "%s\n", sr_last_error());
        return 1;
    }
    printf("slot=%d dis=%u | spch_recg label=%s dis=%u | batch slot=%u dis=%u status=%u frames=%u\n", slot, d1,
           label ? (char *)label : "(null)", d2, res.best_tpl, res.min_dis, res.status, res.frm_num);
    if ((slot >= 0) != (label != NULL) || d1 != d2 || d2 != res.min_dis || (slot >= 0 && (unsigned)slot != res.best_tpl))
        return 1;
    /* (4) what one spch_recg call costs once the engine is warm (plain C, no interpreter in the loop) */
    {
        struct timespec t0, t1;
        uint32_t d3 = 0;
        int i, n = 200;
        for (i = 0; i < 20; i++) (void)spch_recg(VcBuf, &d3);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (i = 0; i < n; i++) (void)spch_recg(VcBuf, &d3);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("spch_recg: %.1f us per call (mean of %d, one call in flight)\n",
               ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / n, n);
        if (d3 != d2) return 1;
        /* ... and the reference's own sequence: noise_atap, VAD, get_mfcc, dtw per slot (main.c:258-291) */
        for (i = 0; i < 20; i++) (void)recognise(VcBuf, store, n_slots, &d3);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (i = 0; i < n; i++) (void)recognise(VcBuf, store, n_slots, &d3);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("noise_atap + VAD + get_mfcc + %u x dtw: %.1f us per capture\n", n_slots,
               ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / n);
        if (d3 != d1) return 1;
    }
    return 0;
}
