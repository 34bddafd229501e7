// sr_dtw_cells.h -- the small-launch DTW kernel (k_dtw_cells.hip): one workgroup per (utterance, template) pair.
#pragma once
#include <vector>

#include "sr_device.h"

namespace sr {

// the most points of dtw_limit's band any (utterance, template) pair of a store can have (0: not worth setting up); the
// engine keeps it with the store and hands it over in DtwArgs::cells_points.  by_len: the engine's cache per template length
// (the frame cap of an engine never changes), so that a store that grows slot by slot -- dtw()'s model cache -- pays per new length
uint32_t dtw_cells_max_points(uint32_t max_frames, const uint32_t *frames, const uint8_t *valid, uint32_t K, std::vector<uint32_t> &by_len);
// the pair's rows and one word per band point fit one workgroup's LDS (and the store has the slack rows the reference's
// do-while reads)
bool dtw_cells_fits(const DtwArgs &a);
size_t dtw_cells_lds(uint32_t max_frames, uint32_t tpl_rows, uint32_t max_points);
// scores[b][k] for every pair, identical to launch_dtw's; meant for launches of a few hundred pairs
void launch_dtw_cells(const DtwArgs &a, hipStream_t s);

}  // namespace sr
