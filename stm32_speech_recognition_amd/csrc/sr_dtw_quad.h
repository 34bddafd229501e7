// sr_dtw_quad.h -- the mid-sized-launch DTW kernel (k_dtw_quad.hip): four lanes per (utterance, template) pair.
#pragma once
#include "sr_device.h"

namespace sr {

// both sequences of at least 1 x 4 pairs fit one workgroup's LDS (feature rows of up to 16 coefficients)
bool dtw_quad_fits(const DtwArgs &a);
// workgroup shape: pu utterances x pk templates (pu * pk <= 64 pairs = 256 lanes) and its LDS bytes
bool dtw_quad_pick(const DtwArgs &a, uint32_t *pu, uint32_t *pk, size_t *lds);
// scores[b][k] for every pair, identical to launch_dtw's; meant for launches of a few thousand to ~100 000 pairs
void launch_dtw_quad(const DtwArgs &a, hipStream_t s);

}  // namespace sr
