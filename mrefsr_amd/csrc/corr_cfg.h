// Constants shared by the correlation pre-filter kernels (corr_prefilter.hip, corr_rowstream.hip), the re-scoring
// kernel and the exact fallback (corr.hip): the workspace layout is a contract between them.
#pragma once

namespace mrefsr_corr {

constexpr int T_PX = 16;            // exact kernel / tile pre-filters: query tile = 8 x 16 pixels ...
constexpr int T_QY = 6, T_QX = 14;  // ... = 6 x 14 patches; tile_flag[] is indexed in this geometry
constexpr int SLOTS = 16;           // candidate slots per query in the global buffer
constexpr int BRUTE_SEG = 32;       // reference segments (blocks) per brute-forced query
constexpr int BRUTE_MAX = 32;       // up to this many overflowed queries are brute-forced one by one; more -> exact kernel on their tiles

struct PrefilterOut {               // views into the caller's workspace (mrefsr_corr_workspace_bytes)
    int *cand_r;                    // [n_pair * P][SLOTS]
    int *cand_n;                    // [n_pair * P]      -1 = overflowed (flagged)
    int *flag_count;                // [1]
    int *flag_list;                 // [n_pair * P]
    int *tile_flag;                 // [n_pair][tiles_y * tiles_x] in the T_QY x T_QX geometry
};

}  // namespace mrefsr_corr
