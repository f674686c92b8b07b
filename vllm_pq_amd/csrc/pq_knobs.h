// Diagnostic knobs of the dispatch (tests, tuning tools): the PQ_AMD_* environment variables, read ONCE when the
// library is first used and again only when pq_debug_reload_knobs() is called -- never per launch.
//   PQ_AMD_FORCE_KERNEL   "generic" | "stream" | "tiled" | "decode" | "wide"
//   PQ_AMD_STREAM_CFG     "NFxTF" tile of the streaming kernel
//   PQ_AMD_STREAM_KSPLIT  blocks per tile along K of the streaming kernel
//   PQ_AMD_TILED_CFG      tile shape index of the tiled kernel
//   PQ_AMD_TILED_SLOTS    blocks per XCD of the tiled kernel
//   PQ_AMD_TILED_SPLIT    the tiled kernel's second launch for a last partial round: 0 never, N > 0 after N row tiles
//   PQ_AMD_DECODE_CFG     "WAVESxNF" block shape of the decode kernel
//   PQ_AMD_WIDE_CFG       "NFxROUNDS" block shape of the mid-batch kernel
#pragma once

namespace pq {

struct Knobs {
    int force_kernel;      // -1: none; else the dispatcher's kernel number
    int stream_nf, stream_tf;   // 0: none
    int stream_ksplit;     // 0: none
    int tiled_cfg;         // -1: none
    int tiled_slots;       // 0: none
    int tiled_split;       // -1: chosen by the launch model; 0: never; N > 0: after N row tiles of 512 rows
    int decode_waves, decode_nf;   // 0: none
    int wide_nf, wide_rounds;      // 0: none
};

const Knobs &knobs();      // thread-safe; the first call reads the environment
void reload_knobs();       // re-reads the environment; not to be called while another thread launches

}  // namespace pq
