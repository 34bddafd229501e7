/*
 * multi_gpu_demo.c -- the recognition path on several MI355X from ONE plain-C process (include/sr_engine.h,
 * "multi-GPU" section): utterances sharded over the devices, templates replicated, one RCCL all-gather of the
 * per-template score matrix (what the firmware's slot scan, Src/APP/main.c:279-291, yields per utterance).
 *
 *   gcc -std=gnu99 -Iinclude examples/multi_gpu_demo.c -Lstm32_speech_recognition_amd -lsr_engine \
 *       -Wl,-rpath,stm32_speech_recognition_amd -o multi_gpu_demo
 *   ./multi_gpu_demo store.bin captures.bin n_devices
 *
 * store.bin    : flash-layout template store, 4096-byte v_ftr_tag slots (Src/BSP/Flash.H:11-20)
 * captures.bin : B capture buffers of 16000 u16 samples (Src/BSP/ADC.H:8-9)
 * Prints, per capture, slot / distance from the sharded run and checks them against a single-device run.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sr_engine.h"

#define VcBuf_Len 16000
#define size_per_ftr 4096

static void *slurp(const char *path, long *sz)
{
    FILE *f = fopen(path, "rb");
    void *p;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    p = malloc((size_t)*sz);
    if (fread(p, 1, (size_t)*sz, f) != (size_t)*sz) p = NULL;
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    long ssz = 0, csz = 0;
    uint8_t *store;
    uint16_t *caps;
    int devices[64], n_dev, i;
    unsigned B, K, b;
    sr_config cfg;
    sr_multi *m = NULL;
    sr_engine *one = NULL;
    sr_result *r_multi, *r_one;
    uint32_t *s_multi, *s_one;

    if (argc < 4) {
        fprintf(stderr, "usage: %s store.bin captures.bin n_devices\n", argv[0]);
        return 2;
    }
    store = slurp(argv[1], &ssz);
    caps = slurp(argv[2], &csz);
    n_dev = atoi(argv[3]);
    if (!store || !caps || n_dev < 1 || n_dev > 64) return 2;
    for (i = 0; i < n_dev; i++) devices[i] = i;
    B = (unsigned)(csz / (2 * VcBuf_Len));
    K = (unsigned)(ssz / size_per_ftr);
    sr_default_config(&cfg);
    if (sr_multi_create(&cfg, devices, (uint32_t)n_dev, &m) != SR_OK ||
        sr_multi_set_templates(m, store, K, size_per_ftr) != SR_OK) {
        fprintf(stderr, "%s\n", sr_last_error());
        return 1;
    }
    r_multi = calloc(B, sizeof *r_multi);
    r_one = calloc(B, sizeof *r_one);
    s_multi = calloc((size_t)B * K, 4);
    s_one = calloc((size_t)B * K, 4);
    if (sr_multi_recognize(m, caps, VcBuf_Len, VcBuf_Len, B, r_multi, s_multi) != SR_OK) {
        fprintf(stderr, "%s\n", sr_last_error());
        return 1;
    }
    cfg.device = 0;
    if (sr_create(&cfg, &one) != SR_OK || sr_set_templates(one, store, K, size_per_ftr) != SR_OK ||
        sr_recognize_batch(one, caps, VcBuf_Len, VcBuf_Len, B, r_one, s_one, NULL, NULL) != SR_OK) {
        fprintf(stderr, "%s\n", sr_last_error());
        return 1;
    }
    for (b = 0; b < B; b++) printf("capture %u: slot %u dis %u status %u\n", b, r_multi[b].best_tpl, r_multi[b].min_dis, r_multi[b].status);
    if (memcmp(r_multi, r_one, B * sizeof *r_one) || memcmp(s_multi, s_one, (size_t)B * K * 4)) {
        printf("MISMATCH between %d-device and single-device results\n", n_dev);
        return 1;
    }
    printf("ok: %u captures x %u slots on %d device(s), gathered scores identical to the single-device run\n", B, K, n_dev);
    sr_destroy(one);
    sr_multi_destroy(m);
    return 0;
}
