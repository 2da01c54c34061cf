// C entry point of scan_fwd: validation + dtype dispatch (kernels live in scan_fwd_impl.h / scan_fwd_<dtype>.hip)
#include "dm_common.h"
namespace dm {
constexpr int SCAN_PF = 8;
constexpr int FWD_CKE = 4;      // checkpoint spacing (scan_fwd_impl.h)
int scan_fwd_f32(const dm_scan_fwd_args& a, hipStream_t st);
int scan_fwd_bf16(const dm_scan_fwd_args& a, hipStream_t st);
int scan_fwd_f16(const dm_scan_fwd_args& a, hipStream_t st);
}  // namespace dm

static dm_scan_fwd_args scan_fwd_normalised(dm_scan_fwd_args a) {
    if (a.flags & DM_FLAG_DELTA_ACTIVATED) {          // delta already holds softplus(raw + bias): the forward uses it as is
        a.flags &= ~(DM_FLAG_DELTA_ACTIVATED | DM_FLAG_DELTA_SOFTPLUS);
        a.delta_bias = nullptr;
    }
    return a;
}

extern "C" int dm_selective_scan_fwd(const dm_scan_fwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_selective_scan_fwd: null args"); return DM_ERR_ARG; }
    const dm_scan_fwd_args a = scan_fwd_normalised(*args);
    if (!a.u || !a.delta || !a.out || !a.A || !a.B || !a.C) {
        set_error("dm_selective_scan_fwd: null tensor pointer"); return DM_ERR_ARG;
    }
    if (a.nseq <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ngroups <= 0) {
        set_error("dm_selective_scan_fwd: non-positive size"); return DM_ERR_ARG;
    }
    if (a.nseq > 65535) { set_error("dm_selective_scan_fwd: nseq %d > 65535", a.nseq); return DM_ERR_ARG; }
    if (a.u_sd != 1 || a.dt_sd != 1 || a.o_sd != 1 || (a.z && a.z_sd != 1) || a.B_sn != 1 || a.C_sn != 1) {
        set_error("dm_selective_scan_fwd: needs token-major tensors (channel stride 1, state stride 1)");
        return DM_ERR_LAYOUT;
    }
    if (a.dim % a.ngroups != 0 || (a.ngroups > 1 && (a.dim / a.ngroups) % WAVE != 0)) {
        set_error("dm_selective_scan_fwd: dim/ngroups must be a multiple of 64"); return DM_ERR_LAYOUT;
    }
    if (a.batch_per_dir > 0 && a.nseq % a.batch_per_dir != 0) {
        set_error("dm_selective_scan_fwd: nseq %% batch_per_dir != 0"); return DM_ERR_ARG;
    }
    if (a.ckpt && a.ckpt_every != FWD_CKE) {
        set_error("dm_selective_scan_fwd: ckpt_every must be %d", FWD_CKE); return DM_ERR_ARG;
    }
    if (a.ckpt && a.ckpt_dtype != (a.io_dtype == DM_BF16 ? DM_BF16 : DM_F32)) {
        set_error("dm_selective_scan_fwd: ckpt_dtype must be DM_BF16 for bf16 I/O and DM_F32 otherwise"); return DM_ERR_DTYPE;
    }
    if ((a.z_row_index == nullptr) != (a.out_row_index == nullptr)) {
        set_error("dm_selective_scan_fwd: z_row_index and out_row_index must both be set or both be NULL"); return DM_ERR_ARG;
    }
    if (a.flags & DM_FLAG_OUT_ACCUMULATE) {
        if (!a.z || !a.z_row_index || a.dstate != 16 || !(a.flags & DM_FLAG_DELTA_SOFTPLUS) || (a.flags & DM_FLAG_A_SHARED) ||
            (a.batch_per_dir > 0 && a.batch_per_dir != a.nseq) || a.last_state) {
            set_error("dm_selective_scan_fwd: DM_FLAG_OUT_ACCUMULATE needs z, row indices, d_state 16, delta softplus, ONE direction per launch, no last_state / A_SHARED");
            return DM_ERR_ARG;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return scan_fwd_f32(a, st);
        case DM_BF16: return scan_fwd_bf16(a, st);
        case DM_F16: return scan_fwd_f16(a, st);
        default: set_error("dm_selective_scan_fwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}

// n congruent launches in one when the shape selects the small-launch (chunk-parallel) kernel, which takes an array of argument
// structs (dm_common.h mix_args); the sequential kernel runs them one after the other.
extern "C" int dm_selective_scan_fwd_n(const dm_scan_fwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0 || n > 16) { set_error("dm_selective_scan_fwd_n: null args / n not in 1..16"); return DM_ERR_ARG; }
    dm_scan_fwd_args norm[16];
    for (int i = 0; i < n; ++i) norm[i] = scan_fwd_normalised(args[i]);
    return mix_launch_n(norm, n, [&](const dm_scan_fwd_args* a) { return dm_selective_scan_fwd(a, stream); },
                        [](const dm_scan_fwd_args& x, const dm_scan_fwd_args& y) {
                            using A = dm_scan_fwd_args;
                            return mix_congruent(x, y, &A::u, &A::delta, &A::z, &A::out, &A::B, &A::C, &A::A, &A::D, &A::delta_bias,
                                                 &A::z_row_index, &A::out_row_index, &A::ckpt, &A::last_state);
                        });
}
