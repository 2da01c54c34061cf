// scan_fwd: f32 I/O instantiations (split per dtype so the library builds in parallel)
#include "scan_fwd_chunked.h"
namespace dm {
int scan_fwd_f32(const dm_scan_fwd_args& a, hipStream_t st) { return dispatch_fwd<float>(a, st); }
}  // namespace dm
