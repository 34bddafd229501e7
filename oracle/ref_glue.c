/*
 * TEST INFRASTRUCTURE ONLY -- glue around the reference's own objects.
 *
 * oracle/_ref/libsr_ref.so = the reference's Src/Speech_Recog/{VAD,MFCC,DTW}.C
 * compiled VERBATIM from /root/reference (never copied) + oracle/q15_fft.c for
 * the assembly FFT + this file.  It is "tier (i)" of the oracle: ground truth
 * at the reference's compile-time constants (8 kHz, 160/80 framing, 1024-pt,
 * 24 Mel, 12 coefficients, <=119 frames).
 *
 * This file supplies what the firmware gets from elsewhere:
 *   - USART1_printf (Src/BSP/USART.C:96-102): debug channel, stubbed to a no-op;
 *   - the recognition driver spch_recg (Src/APP/main.c:249-296), which cannot be
 *     compiled verbatim (main.c pulls in the LCD/touch/flash BSP).  It is
 *     restated here over an explicit template store instead of the absolute
 *     flash window of Src/BSP/Flash.H:11-20; slot stride, save_mask and the
 *     strict-< first-minimum scan are the reference's.
 */
#include "stm32f10x.h"
#include "ADC.h"
#include "VAD.H"
#include "MFCC.H"
#include "DTW.H"
#include <stddef.h>

void USART1_printf(char *fmt, ...) { (void)fmt; }

#define SR_REF_SAVE_MASK 12345u /* Flash.H:11 */

unsigned sr_ref_sizeof_ftr(void) { return (unsigned)sizeof(v_ftr_tag); }
unsigned sr_ref_vv_frm_max(void) { return (unsigned)vv_frm_max; }
unsigned sr_ref_vcbuf_len(void) { return (unsigned)VcBuf_Len; }
unsigned sr_ref_atap_len(void) { return (unsigned)atap_len; }

/* noise_atap + VAD with segment bounds returned as sample offsets (-1 = NULL). */
void sr_ref_vad(const u16 *buf, u16 buf_len, u16 noise_len, atap_tag *atap, s32 *seg /* [2*max_vc_con] */)
{
    valid_tag vv[max_vc_con];
    int i;
    noise_atap(buf, noise_len, atap);
    VAD(buf, buf_len, vv, atap);
    for (i = 0; i < max_vc_con; i++) {
        seg[2 * i + 0] = vv[i].start ? (s32)(vv[i].start - buf) : -1;
        seg[2 * i + 1] = vv[i].end ? (s32)(vv[i].end - buf) : -1;
    }
}

/* get_mfcc on buf[start..end) given as offsets. */
void sr_ref_mfcc(u16 *buf, s32 start, s32 end, atap_tag *atap, v_ftr_tag *ftr)
{
    valid_tag v;
    v.start = buf + start;
    v.end = buf + end;
    get_mfcc(&v, ftr, atap);
}

/*
 * spch_recg (main.c:249-296) over store[n_slots] with byte stride `stride`.
 * Returns 0 ok, 1 VAD fail (main.c:261-266), 2 MFCC fail (main.c:269-274).
 * scores (optional) receives cur_dis of every slot.
 */
int sr_ref_spch_recg_seg(u16 *v_dat, u16 buf_len, u16 noise_len, const u8 *store, u32 n_slots, u32 stride,
                         u32 seg_idx, v_ftr_tag *ftr, u32 *best_slot, u32 *mtch_dis, u32 *scores);

int sr_ref_spch_recg(u16 *v_dat, u16 buf_len, u16 noise_len, const u8 *store, u32 n_slots, u32 stride,
                     v_ftr_tag *ftr, u32 *best_slot, u32 *mtch_dis, u32 *scores)
{
    return sr_ref_spch_recg_seg(v_dat, buf_len, noise_len, store, n_slots, stride, 0, ftr, best_slot, mtch_dis, scores);
}

/* same driver on segment seg_idx of the VAD output (segment 0 = the firmware's behaviour, main.c:268) */
int sr_ref_spch_recg_seg(u16 *v_dat, u16 buf_len, u16 noise_len, const u8 *store, u32 n_slots, u32 stride,
                         u32 seg_idx, v_ftr_tag *ftr, u32 *best_slot, u32 *mtch_dis, u32 *scores)
{
    atap_tag atap;
    valid_tag vv[max_vc_con];
    u32 i, min_dis, min_idx;

    *best_slot = 0;
    noise_atap(v_dat, noise_len, &atap);
    VAD(v_dat, buf_len, vv, &atap);
    if (vv[seg_idx].end == (void *)0) {
        *mtch_dis = dis_err;
        return 1;
    }
    get_mfcc(&vv[seg_idx], ftr, &atap);
    if (ftr->frm_num == 0) {
        *mtch_dis = dis_err;
        return 2;
    }
    min_idx = 0;
    min_dis = dis_max;
    for (i = 0; i < n_slots; i++) {
        v_ftr_tag *mdl = (v_ftr_tag *)(store + (size_t)i * stride);
        u32 cur = (mdl->save_sign == SR_REF_SAVE_MASK) ? dtw(ftr, mdl) : dis_err;
        if (scores)
            scores[i] = cur;
        if (cur < min_dis) {
            min_dis = cur;
            min_idx = i;
        }
    }
    *best_slot = min_idx;
    *mtch_dis = min_dis;
    return 0;
}
