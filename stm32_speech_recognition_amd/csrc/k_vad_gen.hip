// k_vad_gen.hip -- the VAD kernel template (sr_vad_dev.h) for the framings of the GENERIC front end: frame_len = 2 * hop,
// hop a multiple of 8.  No reference counterpart for these constants (the firmware is built for 160 / 80, VAD.H:5-8); the
// algorithm and every integer rule are the reference's, checked against the parametrised oracle.
#include "sr_vad_dev.h"

namespace sr {

bool vad_framing_supported(uint32_t frame_len, uint32_t hop)
{
    if (frame_len != 2 * hop) return false;
    return frame_len == 160 || frame_len == 320 || frame_len == 240 || frame_len == 256 || frame_len == 400 || frame_len == 512;
}

template <int FL>
static void launch_one(const VadArgs &a, hipStream_t s)
{
    const dim3 grid((a.B + kVadWaves - 1) / kVadWaves), block(64 * kVadWaves);
    if (a.atap_in == nullptr) hipLaunchKernelGGL((k_vad<FL, FL / 2, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_vad<FL, FL / 2, false>), grid, block, 0, s, a);
}

void launch_vad_other(const VadArgs &a, hipStream_t s)
{
    switch (a.frame_len) {
    case 240: launch_one<240>(a, s); break;
    case 256: launch_one<256>(a, s); break;
    case 400: launch_one<400>(a, s); break;
    case 512: launch_one<512>(a, s); break;
    default: break;  // sr_create accepts only the framings of vad_framing_supported
    }
}

}  // namespace sr
