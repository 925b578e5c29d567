// rpe_run_ops: a prepared launch list walked on the host side of the C ABI (include/rpe.h, "Prepared launch lists").
// One call enqueues what the reference's Python loops enqueue launch by launch -- RAFT.forward's update iterations
// (core/RAFT/core/raft.py; call sites core/pose/pose_net.py:47,65,129) and the encoders' layers (core/RAFT/core/extractor.py).
// Every op goes through the PUBLIC entry point it names: same argument checks, same kernels, same results as separate calls.
#include "rpe_common.h"

// the ctypes mirrors of robust-pose-estimator_amd/_lib.py were written against these sizes (tests/test_launch_lists_host.py holds them to the same numbers)
static_assert(sizeof(rpe_op) == 16, "rpe_op: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_conv_desc) == 200, "rpe_conv_desc: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_corr_lookup_args) == 48, "rpe_corr_lookup_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_corr_build_args) == 48, "rpe_corr_build_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_stem_conv_args) == 96, "rpe_stem_conv_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_flow_update_args) == 96, "rpe_flow_update_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_copy_planes_args) == 48, "rpe_copy_planes_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_instnorm_finalize_args) == 40, "rpe_instnorm_finalize_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_instnorm_apply_args) == 64, "rpe_instnorm_apply_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_upsample_convex_args) == 40, "rpe_upsample_convex_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_lookup_conv1x1_args) == 96, "rpe_lookup_conv1x1_args: layout changed -- update _lib.py and RPE_ABI_VERSION");
static_assert(sizeof(rpe_solve_opts) == 32, "rpe_solve_opts: layout changed -- update _lib.py and RPE_ABI_VERSION");

template <typename A>
static inline const A* as(const rpe_op& op) { return static_cast<const A*>(op.args); }

static int run_one(const rpe_op& op, void* const* streams, int n_streams) {
    if (op.stream < 0 || op.stream >= n_streams || !op.args) return RPE_E_BADARG;
    void* st = streams[op.stream];
    switch (op.kind) {
    case RPE_OP_CONV_FUSED: return rpe_conv_fused(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV_WINO: return rpe_conv_wino(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV_WINO1D: return rpe_conv_wino1d(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV1X1: return rpe_conv1x1(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV_WINO_X3: return rpe_conv_wino_x3(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV_WINO1D_X3: return rpe_conv_wino1d_x3(as<rpe_conv_desc>(op), st);
    case RPE_OP_CONV1X1_X3: return rpe_conv1x1_x3(as<rpe_conv_desc>(op), st);
    case RPE_OP_CORR_LOOKUP: {
        const auto* a = as<rpe_corr_lookup_args>(op);
        return rpe_corr_lookup(a->pyramid, a->coords, a->b, a->h8, a->w8, a->levels, a->radius, a->out, st);
    }
    case RPE_OP_LOOKUP_CONV1X1: {
        const auto* a = as<rpe_lookup_conv1x1_args>(op);
        return rpe_corr_lookup_conv1x1(a->pyramid, a->coords, a->b, a->h8, a->w8, a->levels, a->radius, a->packed, a->bias, a->relu, a->out,
                                       a->out_batch_stride, a->out2, a->out2_batch_stride, st);
    }
    case RPE_OP_CORR_BUILD: {
        const auto* a = as<rpe_corr_build_args>(op);
        return rpe_corr_build_ex(a->fmap1, a->fmap2, a->b, a->c, a->h8, a->w8, a->levels, a->feature_dtype, a->pyramid, st);
    }
    case RPE_OP_STEM_CONV: {
        const auto* a = as<rpe_stem_conv_args>(op);
        return rpe_stem_conv(a->image, a->b, a->cin, a->h, a->w, a->stride, a->div, a->mul, a->sub, a->packed, a->cout, a->bias, a->scale, a->relu,
                             a->out, a->stats, st);
    }
    case RPE_OP_FLOW_UPDATE: {
        const auto* a = as<rpe_flow_update_args>(op);
        return rpe_conv3x3_to2_flow(a->x, a->weight, a->bias, a->b, a->c, a->h, a->w, a->coords, a->coords_out, a->flow_out, a->dst1,
                                    a->dst1_batch_stride, a->dst2, a->dst2_batch_stride, st);
    }
    case RPE_OP_COPY_PLANES: {
        const auto* a = as<rpe_copy_planes_args>(op);
        return rpe_copy_planes(a->src, a->src_batch_stride, a->dst, a->dst_batch_stride, a->b, a->c, a->hw, st);
    }
    case RPE_OP_INSTNORM_FINALIZE: {
        const auto* a = as<rpe_instnorm_finalize_args>(op);
        return rpe_instnorm_finalize(a->partials, a->tiles, a->b, a->c, a->hw, a->eps, a->mean_inv, st);
    }
    case RPE_OP_INSTNORM_APPLY: {
        const auto* a = as<rpe_instnorm_apply_args>(op);
        return rpe_instnorm_apply_ex(a->x, a->partials, a->tiles, a->b, a->c, a->hw, a->eps, a->relu, a->residual, a->residual_mean_inv, a->out, st);
    }
    case RPE_OP_UPSAMPLE_CONVEX: {
        const auto* a = as<rpe_upsample_convex_args>(op);
        return rpe_upsample_convex(a->flow, a->mask, a->b, a->h8, a->w8, a->out, st);
    }
    case RPE_OP_EVENT_RECORD: {
        hipEvent_t ev = (hipEvent_t) * static_cast<void* const*>(op.args);
        if (!ev) return RPE_OK;
        return hipEventRecord(ev, (hipStream_t)st) == hipSuccess ? RPE_OK : RPE_E_LAUNCH;
    }
    case RPE_OP_STREAM_WAIT: {
        hipEvent_t ev = (hipEvent_t) * static_cast<void* const*>(op.args);
        if (!ev) return RPE_OK;
        return hipStreamWaitEvent((hipStream_t)st, ev, 0) == hipSuccess ? RPE_OK : RPE_E_LAUNCH;
    }
    default: return RPE_E_BADARG;
    }
}

extern "C" int rpe_run_ops(const rpe_op* ops, int n_ops, void* const* streams, int n_streams, int* failed_op) {
    if (failed_op) *failed_op = -1;
    if (!ops || n_ops < 0 || !streams || n_streams <= 0) return RPE_E_BADARG;
    for (int i = 0; i < n_ops; ++i) {
        const int st = run_one(ops[i], streams, n_streams);
        if (st != RPE_OK) {
            if (failed_op) *failed_op = i;
            return st;
        }
    }
    return RPE_OK;
}
