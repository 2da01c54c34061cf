// scan_bwd: f32 I/O instantiations (split per dtype so the library builds in parallel)
#include "scan_bwd_impl.h"
namespace dm {
int scan_bwd_f32(const dm_scan_bwd_args& a, hipStream_t st) { return bwd_dispatch_bc<float>(a, st); }
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

