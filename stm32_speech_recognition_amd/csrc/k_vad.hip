// k_vad.hip -- noise_atap (VAD.C:22-71) + VAD (VAD.C:97-218) + frame count of get_mfcc (MFCC.C:102-107), one wave per capture buffer.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_vad_dev.h"

namespace sr {

// Re-targets the per-utterance records at VAD segment `seg_idx` (0..2): the frame and DTW kernels always work
// on "segment 0" of the record they are given.  Frame count and status follow MFCC.C:102-107 / main.c:261-274.
__global__ void k_select_segment(const sr_vad_rec *in, sr_vad_rec *out, uint32_t B, uint32_t seg_idx, uint32_t max_frames,
                                 uint32_t frame_len, uint32_t hop)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // the chosen segment is read straight from memory (a register copy of the record indexed by seg_idx would live in
    // scratch) and the output record is written by fields
    const int st = in[b].seg[2 * seg_idx], en = in[b].seg[2 * seg_idx + 1];
    uint32_t status, frm_num = 0;
    if (en < 0) {
        status = SR_ST_VAD_FAIL;
    } else if (st < 1) {
        status = SR_ST_SEG_OOB;
    } else {
        const uint32_t n = ((((uint32_t)(en - st) - frame_len) / hop) + 1) & 0xFFFF;
        status = n > max_frames ? SR_ST_MFCC_FAIL : SR_ST_OK;
        frm_num = n > max_frames ? 0 : n;
    }
    const sr_atap atap = in[b].atap;
    const int s2 = in[b].seg[2], s3 = in[b].seg[3], s4 = in[b].seg[4], s5 = in[b].seg[5];
    const uint32_t pad = in[b]._pad;
    sr_vad_rec *o = out + b;  // may alias in + b: everything has been read
    o->atap = atap;
    o->seg[0] = st;
    o->seg[1] = en;
    o->seg[2] = s2;
    o->seg[3] = s3;
    o->seg[4] = s4;
    o->seg[5] = s5;
    o->frm_num = frm_num;
    o->status = status;
    o->_pad = pad;
}
void launch_select_segment(const sr_vad_rec *in, sr_vad_rec *out, uint32_t B, uint32_t seg_idx, uint32_t max_frames,
                           uint32_t frame_len, uint32_t hop, hipStream_t s)
{
    if (!B) return;
    hipLaunchKernelGGL(k_select_segment, dim3((B + 255) / 256), dim3(256), 0, s, in, out, B, seg_idx, max_frames,
                       frame_len, hop);
}

void launch_vad_other(const VadArgs &a, hipStream_t s);

void launch_vad(const VadArgs &a, hipStream_t s)
{
    if (!a.B) return;
    if (a.wide) {
        launch_vad_wide(a, s);
        return;
    }
    const dim3 grid((a.B + kVadWaves - 1) / kVadWaves), block(64 * kVadWaves);
    const bool own_thresholds = a.atap_in == nullptr;  // noise_atap runs in the kernel: mid is a 16-bit quantity
    if (a.frame_len != 160 && a.frame_len != 320) {
        launch_vad_other(a, s);  // k_vad_gen.hip: the other accepted framings
        return;
    }
    if (a.frame_len == 320) {
        if (own_thresholds) hipLaunchKernelGGL((k_vad<320, 160, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_vad<320, 160, false>), grid, block, 0, s, a);
    } else {
        if (own_thresholds) hipLaunchKernelGGL((k_vad<160, 80, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_vad<160, 80, false>), grid, block, 0, s, a);
    }
}

}  // namespace sr
