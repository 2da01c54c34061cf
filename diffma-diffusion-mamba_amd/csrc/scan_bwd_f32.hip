// scan_bwd: f32 I/O instantiations (split per dtype so the library builds in parallel)
#include "scan_bwd_chunked.h"
namespace dm {
int scan_bwd_f32(const dm_scan_bwd_args& a, hipStream_t st) { return bwd_dispatch<float>(a, st); }
}  // namespace dm

extern "C" int dm_scan_bwd_group_channels(int dstate) {
    using namespace dm;
    switch (dstate) {
        case 16: return (WAVE / bwd_split<16>::value) * BWD_WAVES;
#ifndef DM_FAST_BUILD
        case 8: return (WAVE / bwd_split<8>::value) * BWD_WAVES;
        case 32: return (WAVE / bwd_split<32>::value) * BWD_WAVES;
#endif
        default: return -1;
    }
}

// channels per dB/dC partial row for THIS launch: 64 when the chunk-parallel kernel (scan_bwd_chunked.h) will take it,
// dm_scan_bwd_group_channels(dstate) otherwise.  The host sizes dBC_partial [nseq][seqlen][ceil(dim/GC)][2*dstate] with it.
extern "C" int dm_scan_bwd_launch_group_channels(int nseq, int dim, int seqlen, int dstate, int flags) {
    if (dm::bwd_chunked_lc(nseq, dim, seqlen, dstate, flags) > 0) return dm::WAVE;
    return dm_scan_bwd_group_channels(dstate);
}
