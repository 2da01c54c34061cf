// K2  dm_selective_scan_bwd -- placeholder until the reverse-time kernel lands (same commit series).
#include "dm_common.h"
extern "C" int dm_selective_scan_bwd(const dm_scan_bwd_args* args, void* stream) {
    (void)args; (void)stream;
    dm::set_error("dm_selective_scan_bwd: not built yet");
    return DM_ERR_ARG;
}
