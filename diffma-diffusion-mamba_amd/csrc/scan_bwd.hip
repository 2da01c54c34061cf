// C entry point of scan_bwd: validation + dtype dispatch (kernels live in scan_bwd_impl.h / scan_bwd_<dtype>.hip)
#include "dm_common.h"
namespace dm {
constexpr int BWD_CK = 8;
constexpr int BWD_SUB = 4;     // = checkpoint spacing (scan_bwd_impl.h)
int scan_bwd_f32(const dm_scan_bwd_args& a, hipStream_t st);
int scan_bwd_bf16(const dm_scan_bwd_args& a, hipStream_t st);
int scan_bwd_f16(const dm_scan_bwd_args& a, hipStream_t st);
}  // namespace dm

extern "C" int dm_selective_scan_bwd(const dm_scan_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_selective_scan_bwd: null args"); return DM_ERR_ARG; }
    const dm_scan_bwd_args& a = *args;
    if (!a.u || !a.delta || !a.dout || !a.A || !a.B || !a.C || !a.du || !a.ddelta || !a.dBC_partial || !a.dA_partial) {
        set_error("dm_selective_scan_bwd: null tensor pointer"); return DM_ERR_ARG;
    }
    if ((a.z != nullptr) != (a.dz != nullptr)) { set_error("dm_selective_scan_bwd: dz must be given iff z is"); return DM_ERR_ARG; }
    if (a.nseq <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ngroups <= 0) { set_error("dm_selective_scan_bwd: non-positive size"); return DM_ERR_ARG; }
    if (a.nseq > 65535) { set_error("dm_selective_scan_bwd: nseq %d > 65535", a.nseq); return DM_ERR_ARG; }
    if (a.ckpt_every != BWD_SUB) { set_error("dm_selective_scan_bwd: ckpt_every must be %d", BWD_SUB); return DM_ERR_ARG; }
    if (!a.ckpt) { set_error("dm_selective_scan_bwd: ckpt (the checkpoints written by dm_selective_scan_fwd) is required"); return DM_ERR_ARG; }
    if (a.ckpt && a.ckpt_dtype != (a.io_dtype == DM_BF16 ? DM_BF16 : DM_F32)) {
        set_error("dm_selective_scan_bwd: ckpt_dtype must be DM_BF16 for bf16 I/O and DM_F32 otherwise"); return DM_ERR_DTYPE;
    }
    if (a.u_sd != 1 || a.dt_sd != 1 || a.do_sd != 1 || a.du_sd != 1 || a.ddt_sd != 1 || (a.z && (a.z_sd != 1 || a.dz_sd != 1)) ||
        a.B_sn != 1 || a.C_sn != 1) {
        set_error("dm_selective_scan_bwd: needs token-major tensors (channel stride 1, state stride 1)"); return DM_ERR_LAYOUT;
    }
    if (a.dim % a.ngroups != 0 || (a.ngroups > 1 && (a.dim / a.ngroups) % WAVE != 0)) {
        set_error("dm_selective_scan_bwd: dim/ngroups must be a multiple of 64"); return DM_ERR_LAYOUT;
    }
    if (a.batch_per_dir > 0 && a.nseq % a.batch_per_dir != 0) { set_error("dm_selective_scan_bwd: nseq %% batch_per_dir != 0"); return DM_ERR_ARG; }
    if ((a.flags & DM_FLAG_DELTA_ACTIVATED) && (a.z || !a.z_row_index || a.dstate != 16 || (a.flags & (DM_FLAG_A_SHARED | DM_FLAG_DELTA_SOFTPLUS)))) {
        set_error("dm_selective_scan_bwd: DM_FLAG_DELTA_ACTIVATED is built for the mixer's call pattern (no z, row indices, d_state 16, no A_SHARED / DELTA_SOFTPLUS)");
        return DM_ERR_ARG;
    }
    if ((a.z_row_index == nullptr) != (a.out_row_index == nullptr)) {
        set_error("dm_selective_scan_bwd: z_row_index and out_row_index must both be set or both be NULL"); return DM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return scan_bwd_f32(a, st);
        case DM_BF16: return scan_bwd_bf16(a, st);
        case DM_F16: return scan_bwd_f16(a, st);
        default: set_error("dm_selective_scan_bwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}

// n congruent launches in one when the shape selects the small-launch (chunk-parallel) kernel (dm_common.h mix_args)
extern "C" int dm_selective_scan_bwd_n(const dm_scan_bwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_selective_scan_bwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_scan_bwd_args* a) { return dm_selective_scan_bwd(a, stream); },
                        [](const dm_scan_bwd_args& x, const dm_scan_bwd_args& y) {
                            using A = dm_scan_bwd_args;
                            return mix_congruent(x, y, &A::u, &A::delta, &A::z, &A::dout, &A::B, &A::C, &A::A, &A::D, &A::delta_bias,
                                                 &A::z_row_index, &A::out_row_index, &A::ckpt, &A::du, &A::ddelta, &A::dz,
                                                 &A::dBC_partial, &A::dA_partial, &A::dD_partial, &A::dbias_partial);
                        });
}
