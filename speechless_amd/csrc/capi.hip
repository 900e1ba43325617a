// capi.hip -- extern "C" boundary (include/speechless_hip.h): argument validation + dispatch to the gfx950 kernels.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void sl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" int sl_version(void) { return SL_VERSION; }

// ---- measurement hook: see common.h / include/speechless_hip.h
static hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
extern "C" int sl_profile_next_kernel(void* start_event, void* stop_event) {
    g_prof_start = (hipEvent_t)start_event;
    g_prof_stop = (hipEvent_t)stop_event;
    return SL_OK;
}
void sl_prof_take(hipEvent_t* start, hipEvent_t* stop) {
    *start = g_prof_start;
    *stop = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
}
extern "C" const char* sl_last_error(void) { return g_last_error; }

static int check_geom(const sl_conv_geom* g, const char* who, int cin_mult, int cout_mult) {
    SL_CHECK_ARG(g != nullptr, "%s: geom is null", who);
    SL_CHECK_ARG(g->batch > 0 && g->t_out > 0 && g->taps > 0, "%s: batch/t_out/taps must be positive", who);
    SL_CHECK_ARG(g->cin > 0 && g->cin % cin_mult == 0, "%s: cin=%d must be a positive multiple of %d", who, g->cin,
                 cin_mult);
    SL_CHECK_ARG(g->cout > 0 && g->cout % cout_mult == 0, "%s: cout=%d must be a positive multiple of %d", who,
                 g->cout, cout_mult);
    SL_CHECK_ARG(g->x_row0 >= 0 && g->y_row0 >= 0, "%s: negative row offset", who);
    SL_CHECK_ARG(g->x_row_stride >= g->cin && g->x_row_stride % 8 == 0, "%s: bad x_row_stride", who);
    SL_CHECK_ARG(g->y_row_stride >= g->cout && g->y_row_stride % 8 == 0, "%s: bad y_row_stride", who);
    return SL_OK;
}

extern "C" size_t sl_conv1d_nt_workspace_bytes(const sl_conv_geom* geom, int dtype, int cfg) {
    if (!geom || (dtype != SL_BF16 && dtype != SL_F16) || geom->taps <= 0 || geom->cin <= 0 || geom->cin % 64 || geom->cout <= 0 ||
        geom->cout % 128 || geom->batch <= 0 || geom->t_out <= 0)
        return 0;
    return dtype == SL_F16 ? conv_nt_f16_workspace_bytes(geom, cfg) : conv_nt_bf16_workspace_bytes(geom, cfg);
}

extern "C" int sl_conv1d_nt(const void* x, const void* w, const float* bias, const void* mask, void* y,
                            const sl_conv_geom* geom, int epilogue, int dtype, int out_f32, int cfg, void* workspace,
                            size_t workspace_bytes, void* stream) {
    int rc = check_geom(geom, "sl_conv1d_nt", 64, 128);
    if (rc != SL_OK) return rc;
    SL_CHECK_ARG(x && w && y, "sl_conv1d_nt: null tensor pointer");
    SL_CHECK_ARG(epilogue >= SL_EPI_NONE && epilogue <= SL_EPI_ELU_MASK, "sl_conv1d_nt: unknown epilogue %d", epilogue);
    if (epilogue == SL_EPI_BIAS || epilogue == SL_EPI_BIAS_RELU || epilogue == SL_EPI_BIAS_ELU)
        SL_CHECK_ARG(bias, "sl_conv1d_nt: bias is null");
    if (epilogue == SL_EPI_RELU_MASK || epilogue == SL_EPI_ELU_MASK) SL_CHECK_ARG(mask, "sl_conv1d_nt: mask is null");
    if (dtype == SL_BF16)
        return conv_nt_bf16(x, w, bias, mask, y, geom, epilogue, out_f32, cfg, workspace, workspace_bytes,
                            (hipStream_t)stream);
    if (dtype == SL_F16) {
        SL_CHECK_ARG(out_f32 == 1 || out_f32 == 2, "sl_conv1d_nt(f16): fp32 or plane outputs only (out_f32 = 1 or 2)");
        return conv_nt_f16(x, w, bias, mask, y, geom, epilogue, out_f32, cfg, workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (dtype == SL_F32) return conv_nt_f32(x, w, bias, mask, y, geom, epilogue, cfg, (hipStream_t)stream);
    sl_set_error("sl_conv1d_nt: unknown dtype %d", dtype);
    return SL_ERR_INVALID_ARGUMENT;
}

bool output_softmax_supported(const sl_conv_geom* g, int k);
int output_softmax_bf16(const void* x, const void* w, const float* bias, float* probs, float* logq, float* logits,
                        const sl_conv_geom* g, int k, int logit_stride, long logit_batch_stride, float eps, hipStream_t s);

int output_softmax_select(int variant);
extern "C" int sl_output_softmax_select(int variant) {
    SL_CHECK_ARG(variant >= 0 && variant <= 2, "sl_output_softmax_select: variant %d outside 0..2", variant);
    return output_softmax_select(variant);
}

extern "C" int sl_output_softmax_supported(const sl_conv_geom* geom, int k, int dtype) {
    if (!geom || dtype != SL_BF16 || geom->batch <= 0 || geom->t_out <= 0 || geom->cin <= 0) return 0;
    return output_softmax_supported(geom, k) ? 1 : 0;
}

extern "C" int sl_output_softmax(const void* x, const void* w, const float* bias, float* probs, float* logq, float* logits,
                                 const sl_conv_geom* geom, int k, int logit_stride, int64_t logit_batch_stride, float eps,
                                 int dtype, void* stream) {
    SL_CHECK_ARG(geom && x && w && bias && probs && logq, "sl_output_softmax: null pointer");
    SL_CHECK_ARG(sl_output_softmax_supported(geom, k, dtype),
                 "sl_output_softmax: needs bf16, a 1x1 layer, k <= 32 classes, cin a multiple of 64 with 32 weight rows in LDS");
    SL_CHECK_ARG(geom->x_row0 >= 0 && geom->x_row_stride >= geom->cin && geom->x_row_stride % 8 == 0,
                 "sl_output_softmax: bad input geometry");
    return output_softmax_bf16(x, w, bias, probs, logq, logits, geom, k, logit_stride, (long)logit_batch_stride, eps,
                               (hipStream_t)stream);
}

int conv_chain_select(int rows);
extern "C" int sl_conv1d_chain_select(int tile_rows) {
    SL_CHECK_ARG(tile_rows == 0 || tile_rows == 48 || tile_rows == 64, "sl_conv1d_chain_select: tile_rows must be 0, 48 or 64");
    return conv_chain_select(tile_rows);
}

extern "C" int sl_conv1d_chain_supported(const sl_conv_geom* geom, int n_layers, int dtype) {
    return geom != nullptr && dtype == SL_BF16 && geom->batch > 0 && geom->t_out > 0 && conv_chain_bf16_supported(geom, n_layers);
}

extern "C" int sl_conv1d_chain(const void* x, void* const* ys, const void* const* ws, const float* const* biases,
                               const void* const* masks, const sl_conv_geom* geom, int n_layers, int epilogue, int dtype,
                               void* stream) {
    SL_CHECK_ARG(x && ys && ws && geom, "sl_conv1d_chain: null pointer");
    SL_CHECK_ARG(epilogue == SL_EPI_BIAS_RELU || epilogue == SL_EPI_RELU_MASK,
                 "sl_conv1d_chain: epilogue must be SL_EPI_BIAS_RELU (forward) or SL_EPI_RELU_MASK (input gradient)");
    SL_CHECK_ARG(epilogue == SL_EPI_BIAS_RELU ? biases != nullptr : masks != nullptr,
                 "sl_conv1d_chain: the epilogue's per-layer operand (biases / masks) is null");
    if (!sl_conv1d_chain_supported(geom, n_layers, dtype)) {
        sl_set_error("sl_conv1d_chain: this run does not fit the fused kernel (256 padded channels, odd taps <= 9, 2..8 "
                     "layers, bf16)");
        return SL_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < n_layers; ++i)
        SL_CHECK_ARG(ys[i] && ws[i] && (epilogue == SL_EPI_BIAS_RELU ? (const void*)biases[i] : masks[i]),
                     "sl_conv1d_chain: null pointer for layer %d", i);
    return conv_chain_bf16(x, ys, ws, biases, masks, geom, n_layers, epilogue, (hipStream_t)stream);
}

extern "C" size_t sl_conv1d_wgrad_workspace_bytes(const sl_conv_geom* geom, int dtype, int cfg) {
    if (!geom || geom->taps <= 0 || geom->cin <= 0 || geom->cout <= 0 || geom->batch <= 0) return 0;
    if (dtype == SL_BF16 || dtype == SL_F16) {
        if (geom->cin % 128 || geom->cout % 128) return 0;
        return dtype == SL_F16 ? wgrad_tn_f16_workspace_bytes(geom, cfg, 1) : wgrad_tn_bf16_workspace_bytes(geom, cfg, 1);
    }
    if (geom->cin % 64 || geom->cout % 64) return 0;
    const int splits = wgrad_split_count(geom, wgrad_f32_tile(geom, cfg));
    if (splits <= 1) return 0;
    return (size_t)splits * geom->taps * geom->cin * geom->cout * sizeof(float);
}

extern "C" int sl_conv1d_wgrad(const void* x, const void* g, float* dw, const sl_conv_geom* geom, int dtype, int cfg,
                               void* workspace, size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32 || dtype == SL_F16, "sl_conv1d_wgrad: unknown dtype %d", dtype);
    const int tile = dtype == SL_F32 ? 64 : 128;
    int rc = check_geom(geom, "sl_conv1d_wgrad", tile, tile);
    if (rc != SL_OK) return rc;
    SL_CHECK_ARG(x && g && dw, "sl_conv1d_wgrad: null tensor pointer");
    if (dtype == SL_BF16)
        return wgrad_tn_bf16(x, g, dw, geom, cfg, 1, 0, 0, 0, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == SL_F16)
        return wgrad_tn_f16(x, g, dw, geom, cfg, 1, 0, 0, 0, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    const int splits = wgrad_split_count(geom, wgrad_f32_tile(geom, cfg));
    const size_t need = sl_conv1d_wgrad_workspace_bytes(geom, dtype, cfg);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
        sl_set_error("sl_conv1d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    return wgrad_tn_f32(x, g, dw, geom, (float*)workspace, splits, cfg, (hipStream_t)stream);
}

extern "C" size_t sl_conv1d_wgrad_multi_workspace_bytes(const sl_wgrad_job* jobs, int n_jobs, int dtype) {
    if (!jobs || (dtype != SL_BF16 && dtype != SL_F16)) return 0;
    return dtype == SL_F16 ? wgrad_multi_f16_workspace_bytes(jobs, n_jobs) : wgrad_multi_bf16_workspace_bytes(jobs, n_jobs);
}

extern "C" int sl_conv1d_wgrad_multi(const sl_wgrad_job* jobs, int n_jobs, int dtype, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(jobs != nullptr, "sl_conv1d_wgrad_multi: jobs is null");
    if (dtype == SL_F16) return wgrad_multi_f16(jobs, n_jobs, workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype != SL_BF16) {
        sl_set_error("sl_conv1d_wgrad_multi: bf16 / f16 only");
        return SL_ERR_UNSUPPORTED;
    }
    return wgrad_multi_bf16(jobs, n_jobs, workspace, workspace_bytes, (hipStream_t)stream);
}

// per THREAD: the engine brackets the launches it wants planned for fewer CUs on the thread that enqueues them; a stager
// thread or another engine's thread of the same process never sees the setting
static thread_local int g_available_cus = 256;
int sl_cus() { return g_available_cus; }
extern "C" int sl_set_available_cus(int cus) {
    SL_CHECK_ARG(cus == 0 || (cus >= 64 && cus <= 256), "sl_set_available_cus: 0 (all) or 64 .. 256");
    g_available_cus = cus == 0 ? 256 : cus;
    return SL_OK;
}

extern "C" int sl_conv1d_backward_1x1_supported(const sl_conv_geom* geom, int k_real, int dtype) {
    return geom != nullptr && dtype == SL_BF16 && conv1x1_bwd_bf16_supported(geom, k_real) ? 1 : 0;
}

extern "C" size_t sl_conv1d_backward_1x1_workspace_bytes(const sl_conv_geom* geom, int k_real, int dtype, int cfg) {
    if (!sl_conv1d_backward_1x1_supported(geom, k_real, dtype)) return 0;
    return conv1x1_bwd_bf16_workspace_bytes(geom, cfg);
}

extern "C" int sl_conv1d_backward_1x1(const void* x, const void* g, const void* w_dgrad, void* dx, float* dw,
                                      const sl_conv_geom* geom, int epilogue, int k_real, int dtype, int cfg,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return sl_conv1d_backward_1x1_part(x, g, w_dgrad, dx, dw, geom, epilogue, k_real, dtype, cfg, 0, workspace, workspace_bytes,
                                       stream);
}

extern "C" int sl_conv1d_backward_1x1_part(const void* x, const void* g, const void* w_dgrad, void* dx, float* dw,
                                           const sl_conv_geom* geom, int epilogue, int k_real, int dtype, int cfg,
                                           int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(x && g && w_dgrad && dx && dw && geom, "sl_conv1d_backward_1x1: null pointer");
    SL_CHECK_ARG(epilogue == SL_EPI_RELU_MASK || epilogue == SL_EPI_ELU_MASK,
                 "sl_conv1d_backward_1x1: epilogue must be SL_EPI_RELU_MASK or SL_EPI_ELU_MASK");
    if (!sl_conv1d_backward_1x1_supported(geom, k_real, dtype)) {
        sl_set_error("sl_conv1d_backward_1x1: needs bf16, a 1x1 layer with <= 32 real output channels and a multiple of 128 "
                     "input channels");
        return SL_ERR_UNSUPPORTED;
    }
    SL_CHECK_ARG(geom->x_row0 >= 0 && geom->y_row0 >= 0 && geom->x_row_stride >= geom->cin && geom->y_row_stride >= 32,
                 "sl_conv1d_backward_1x1: bad geometry");
    SL_CHECK_ARG(cfg >= 0, "sl_conv1d_backward_1x1: cfg must be >= 0");
    return conv1x1_bwd_bf16(x, g, w_dgrad, dx, dw, geom, epilogue, cfg, accumulate ? 1 : 0, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

extern "C" size_t sl_conv1d_wgrad_grouped_workspace_bytes(const sl_conv_geom* geom, int groups, int cfg) {
    if (!geom || groups < 1 || geom->taps <= 0 || geom->cin <= 0 || geom->cout <= 0 || geom->batch <= 0) return 0;
    if (geom->cin % 128 || geom->cout % 128) return 0;
    return wgrad_tn_bf16_workspace_bytes(geom, cfg, groups);
}

extern "C" int sl_conv1d_wgrad_grouped(const void* x, const void* g, float* dw, const sl_conv_geom* geom, int groups,
                                       int64_t x_group_stride, int64_t g_group_stride, int64_t dw_group_stride, int cfg,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_geom(geom, "sl_conv1d_wgrad_grouped", 128, 128);
    if (rc != SL_OK) return rc;
    SL_CHECK_ARG(x && g && dw && groups >= 1, "sl_conv1d_wgrad_grouped: null tensor pointer or groups < 1");
    SL_CHECK_ARG(dw_group_stride % 4 == 0, "sl_conv1d_wgrad_grouped: dw_group_stride must be a multiple of 4 floats");
    return wgrad_tn_bf16(x, g, dw, geom, cfg, groups, (long)x_group_stride, (long)g_group_stride,
                         (long)dw_group_stride, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
