// scan_bwd: f16 I/O instantiations (split per dtype so the library builds in parallel)
#include "scan_bwd_chunked.h"
namespace dm {
int scan_bwd_f16(const dm_scan_bwd_args& a, hipStream_t st) { return bwd_dispatch<f16_t>(a, st); }
}  // namespace dm
