// scan_fwd: f16 I/O instantiations (split per dtype so the library builds in parallel)
#include "scan_fwd_chunked.h"
namespace dm {
int scan_fwd_f16(const dm_scan_fwd_args& a, hipStream_t st) { return dispatch_fwd<f16_t>(a, st); }
}  // namespace dm
