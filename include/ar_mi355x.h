/*
 * ar_mi355x.h -- C ABI of the MI355X (gfx950 / CDNA4) implementation of AutoRound's block-tuning hot path.
 *
 * This is the drop-in boundary: plain pointers + sizes + a hipStream_t, no torch types.  Every entry point
 *   - takes DEVICE pointers (HBM resident; 16-byte aligned for the bulk arrays),
 *   - enqueues work on `stream` and returns immediately (never synchronises, never allocates),
 *   - returns 0 on success or the hipError_t of the failed launch (ar_error_string() decodes it);
 *     AR_ERR_UNSUPPORTED (-1) means the argument combination is outside what this build implements.
 *
 * Each function names the reference interface it replaces (paths relative to the intel/auto-round tree).
 * The Python host layer (auto_round_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add to call them from auto_round's own plugin points.
 *
 * Layout conventions (identical to the reference's group-reshaped tensors, data_type/utils.py:29-71):
 *   W, Wq, dWq : [n_groups * gs] row-major == the [out, in] weight viewed as [out*in/gs, gs]  (in % gs == 0)
 *   V          : [n_groups * gs] fp32 rounding offsets (WrapperLinear.value)
 *   wmin, wmax : [n_groups] in the weight dtype, min clamped to <=0 / max clamped to >=0 (wrapper.py:154-164)
 *   min_s,max_s: [n_groups] fp32 (WrapperLinear.min_scale / max_scale)
 * A whole transformer block may be passed as ONE call: the per-layer arrays are slices of block-wide flat
 * buffers, so n_groups is the block total (grouped launch; no per-layer descriptor table is needed).
 */
#ifndef AR_MI355X_H
#define AR_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ar_stream_t; /* hipStream_t */

enum { AR_DT_BF16 = 0, AR_DT_F16 = 1, AR_DT_F32 = 2 };
enum { AR_OK = 0, AR_ERR_UNSUPPORTED = -1 };

/* ABI version of this header; bump on any signature change.  ar_abi_version() of the loaded library must equal it. */
#define AR_ABI_VERSION 27
int ar_abi_version(void);
/* Human-readable text for a non-zero return code of any function below. */
const char* ar_error_string(int code);

/* ---- weight statistics --------------------------------------------------------------------------------------
 * replaces: WrapperLinear._init_tuning_params_and_quant_func weight_min/weight_max (auto_round/wrapper.py:154-164)
 * wmin[g] = min(min_k W[g,k], 0), wmax[g] = max(max_k W[g,k], 0)  -- wavefront shuffle reductions. */
int ar_group_minmax(const void* W, void* wmin, void* wmax, int64_t n_groups, int gs, int w_dt, ar_stream_t stream);

/* absmax[g] = max_k |W[g,k]| (fp32 out); *tensor_absmax (optional, fp32, must be zeroed by the caller) receives the
 * global max via atomic max.  replaces: torch.max(torch.abs(tensor)) in quant_mx (data_type/mxfp.py:268) and
 * calculate_gparam (data_type/nvfp.py:56-64). */
int ar_group_absmax(const void* W, float* absmax, float* tensor_absmax, int64_t n_groups, int gs, int w_dt,
                    ar_stream_t stream);

/* ---- INT fake-quant forward (W2/W3/W4/W8; sym "full range" and asym) -----------------------------------------
 * replaces: quant_tensor_sym / quant_tensor_asym (auto_round/data_type/int.py:165-238, :241-298) as called by
 *           WrapperLinear._qdq_weight (auto_round/wrapper.py:244-293), including its in-place [lo,hi] clamp of
 *           min_scale/max_scale (wrapper.py:257-259; applied on read, see ar_qdq_int_bwd_sgd for the write-back).
 * V, min_s, max_s may be NULL (treated as 0 / 1 / 1 == plain RTN).  scale_out [n_groups] (s_dt) and zp_out
 * [n_groups] (fp32; sym writes 2^(bits-1)) are optional (NULL during tuning, set for the final unwrap call).
 * sym: 0 = asym, 1 = sym "full range", 2 = sym with a searched init scale (AR_SYM_INIT, the algorithm extension's
 * quant_tensor_sym(init_scale=...) branch, int.py:201-216): `wmax` then holds init_scale[n_groups] in the weight dtype,
 * `wmin` is ignored, scale = s_dt(init_scale * max_scale), and the backward yields d min_scale = 0. */
#define AR_SYM_INIT 2
int ar_qdq_int_fwd(const void* W, const float* V, const void* wmin, const void* wmax, const float* min_s,
                   const float* max_s, void* Wq, void* scale_out, float* zp_out, int64_t n_groups, int gs, int bits,
                   int sym, int w_dt, int s_dt, float q_thresh, float lo_bound, float hi_bound, ar_stream_t stream);

/* ---- INT fake-quant backward (unfused; for step-level parity tests) ------------------------------------------
 * replaces: torch autograd through quant_tensor_sym / quant_tensor_asym (auto_round/data_type/int.py:165-298; SURVEY
 * 8-a12, App. A.3).  dV [n_groups*gs], dmin/dmax
 * [n_groups] fp32; any of the three may be NULL. */
int ar_qdq_int_bwd(const void* dWq, const void* W, const float* V, const void* wmin, const void* wmax,
                   const float* min_s, const float* max_s, float* dV, float* dmin, float* dmax, int64_t n_groups,
                   int gs, int bits, int sym, int w_dt, int s_dt, float q_thresh, float lo_bound, float hi_bound,
                   ar_stream_t stream);

/* ---- sign-SGD (unfused) --------------------------------------------------------------------------------------
 * replaces: SignSGD.step -> _single_tensor_sgd: param.add_(sign(g), alpha=-lr)
 *           (auto_round/algorithms/quantization/sign_round/sign_sgd.py:356-389).  lr is read from device memory
 * (one fp32) so that a captured hipGraph can be replayed with a new learning rate. */
int ar_sign_sgd(float* p, const float* g, int64_t n, const float* lr_dev, ar_stream_t stream);

/* ---- fused backward + sign-SGD (+ best-parameter snapshot, + next forward) -----------------------------------
 * One pass over the block: reads dWq, W, V; writes V (and min_s/max_s, clamped-then-stepped); dV never exists in
 * HBM.  replaces, per iteration: autograd backward of the qdq, optimizer.step(), zero_grad()
 * (sign_round/quantizer.py:502,523,805-821) and, when snapshot_flag != NULL and *snapshot_flag != 0, the
 * collect_best_params deep copy (compressors/utils.py:205-217): best_* receive the PRE-update values (V and the
 * clamped scales), i.e. the parameters that produced the loss that was just found to be the best.
 * lr_v_dev / lr_mm_dev: device fp32 learning rates for V and for min/max scales (LinearLR runs on the host).
 * tune_minmax == 0 leaves min_s/max_s untouched (enable_minmax_tuning=False).
 * Wq_next (optional): when non-NULL the kernel also emits the NEXT iteration's fake-quant weight from the updated
 * parameters (same arithmetic as ar_qdq_int_fwd), saving that kernel's read of W and V. */
int ar_qdq_int_bwd_sgd(const void* dWq, const void* W, float* V, const void* wmin, const void* wmax, float* min_s,
                       float* max_s, int64_t n_groups, int gs, int bits, int sym, int w_dt, int s_dt, float q_thresh,
                       float lo_bound, float hi_bound, const float* lr_v_dev, const float* lr_mm_dev, int tune_minmax,
                       const int32_t* snapshot_flag, float* best_V, float* best_min, float* best_max, void* Wq_next,
                       ar_stream_t stream);

/* ---- MSE output loss, forward + backward in one pass ---------------------------------------------------------
 * replaces: _get_loss (MSELoss on fp32 casts, sign_round/quantizer.py:127-158) + (loss*1000).backward()
 *           (:789-803) up to the gradient w.r.t. the block output.
 * dpred (act dtype) = ((2/n) * (pred-ref)) * grad_scale ; *loss_out = mean((pred-ref)^2) ;
 * if loss_accum != NULL: *loss_accum += mean * accum_scale  (the loop's `total_loss += loss.item()/num_elm`
 * without the host sync).  workspace: >= ar_mse_workspace_bytes() bytes of device scratch.
 * token_mask (optional, uint8 [n / row_len]): the valid-token loss mask of the reference (positions whose token id was
 * marked -100 by the calibrator: pads and the last token of every sample; quantization/base.py:257-280,
 * sign_round/quantizer.py:142-151): masked rows contribute 0 to the (still n-normalised) mean and get a zero gradient. */
int64_t ar_mse_workspace_bytes(void);
int ar_mse_loss_fwd_bwd(const void* pred, const void* ref, void* dpred, float* loss_out, float* loss_accum,
                        float accum_scale, int64_t n, int act_dt, float grad_scale, const uint8_t* token_mask,
                        int64_t row_len, void* workspace, ar_stream_t stream);

/* ---- dynamic INT activation fake-quant (W4A8 / W8A8-style schemes), sym = 1 symmetric / 0 asymmetric ---------------
 * replaces: quant_tensor_sym / quant_tensor_asym (auto_round/data_type/int.py:165-238, :241-298) as called by WrapperLinear._qdq_act
 *           (auto_round/wrapper.py:295-321, forward :530-540) with v = 0, tensor_min/max = None and the wrapper's
 *           non-tunable act_min_scale / act_max_scale (= 1), plus the autograd backward w.r.t. the activation.
 * X / Xq / dXq / dX: [n_groups * gs] in a_dt, groups of gs consecutive elements along the hidden dimension
 * (gs % 8 == 0; per-token quantisation = gs equal to the hidden size).  scale_out [n_groups] in s_dt, optional.
 * The gradient includes the path through the dynamic scale: it is routed to the first arg-min / arg-max element of
 * each group exactly as torch's min/max backward does. */
int ar_qdq_int_act_fwd(const void* X, void* Xq, void* scale_out, int64_t n_groups, int gs, int bits, int sym, int a_dt,
                       int s_dt, float q_thresh, ar_stream_t stream);
int ar_int_act_bwd(const void* dXq, const void* X, void* dX, int64_t n_groups, int gs, int bits, int sym, int a_dt, int s_dt,
                   float q_thresh, ar_stream_t stream);

/* int-sym init-scale search of the algorithm extension.
 * replaces: search_scales (auto_round/data_type/int.py:24-86) and the threshold clamp around it in
 *           _resolve_optimized_dtype_funcs.search_int (auto_round/data_type/utils.py:203-209), as called from
 *           SignRoundOptimizedWrapperLinear._init_tuning_params_and_quant_func (sign_roundv2/quantizer.py:104-126).
 * X [n_groups*gs] in x_dt (grouped, padded weight); qw_row [groups_per_row*gs] fp32 per-input-channel importance
 * (the imatrix, pad = 1e-5) or NULL (== 1).  candidates_dev: fp32 numerators c (scale_c = 1/(-c/gmax)), candidate 0
 * must be 2^(bits-1); a later candidate wins only with a strictly smaller loss.  out_raw / out_init [n_groups] in
 * x_dt: the selected scale before / after the signed q_thresh clamp (either may be NULL).  gs % 8 == 0. */
int ar_search_int_scale(const void* X, const float* qw_row, int64_t groups_per_row, const float* candidates_dev,
                        int n_candidates, void* out_raw, void* out_init, int64_t n_groups, int gs, int bits, int x_dt,
                        float q_thresh, ar_stream_t stream);

/* outlier-suppressed MSE loss of the algorithm extension (used when bits < 4 or act_bits <= 4):
 * replaces: SignRoundV2Quantizer._get_loss (auto_round/algorithms/quantization/sign_roundv2/quantizer.py:362-399):
 * the `topk` = max(1, n/1000) largest |pred - ref| (ranked on the activation-dtype difference, like torch.topk on the bf16
 * tensor) are dropped from the loss and get a zero gradient; the rest is the n-normalised MSE above (token mask included).
 * Selection is a two-level radix select on the 15-bit magnitude pattern of the 16-bit difference.  Exactly `topk`
 * elements are dropped: everything above the k-th value plus the LOWEST-INDEX elements tied with it (torch.topk picks an
 * unspecified subset of the ties; the tied values are equal in the ranking dtype, so the loss agrees to fp32 rounding).
 * 16-bit dtypes only, n % 8 == 0. */
int64_t ar_outlier_loss_workspace_bytes(void);
int ar_outlier_mse_loss_fwd_bwd(const void* pred, const void* ref, void* dpred, float* loss_out, float* loss_accum,
                                float accum_scale, int64_t n, int act_dt, float grad_scale, const uint8_t* token_mask,
                                int64_t row_len, int64_t topk, void* workspace, ar_stream_t stream);

/* ---- best-loss bookkeeping on the device ---------------------------------------------------------------------
 * replaces: `if total_loss < best_loss: best_loss = total_loss; last_best_iter = i` (sign_round/quantizer.py:508-517)
 * state[0]=best_loss (init FLT_MAX) state[1]=init_loss state[2]=last total_loss ; istate[0]=snapshot flag
 * istate[1]=last_best_iter istate[2]=number of improvements (0 => the loss was never finite: keep RTN).
 * Also zeroes *total_loss for the next iteration.
 * iter_dev (optional): the iteration number is read from *iter_dev instead of `iter` and *iter_dev is incremented afterwards --
 * the form a captured hipGraph of one iteration needs.  loss_hist (optional, fp32 [iters]): loss_hist[iteration] = total loss,
 * i.e. the per-iteration loss the reference only logs (`loss.item()` per batch, sign_round/quantizer.py:496) without a host
 * synchronisation. */
int ar_best_loss_update(float* total_loss, float* state, int32_t* istate, int32_t iter, int32_t* iter_dev, float* loss_hist,
                        ar_stream_t stream);

/* ---- start of a tuning iteration replayed from a hipGraph ------------------------------------------------------
 * replaces: the host side of one loop iteration of SignRoundQuantizer.quantize_block -- `index_sampler.next_batch()`
 *           (sign_round/quantizer.py:475, compressors/utils.py:388-438) and the learning rates `lr_schedule.step()` left in the
 *           optimizer's param groups (sign_round/quantizer.py:805-821) -- for a loop whose iterations are ONE captured hipGraph:
 * with it = *iter_dev:  cur_idx[j] = sched[it * batch + j] (j < batch);  lr_out[k] = lr_table[k * iters + it] (k < n_lr).
 * The whole index schedule and the scheduler's learning-rate sequence are drawn / computed by the host before the loop and
 * uploaded once; ar_best_loss_update(iter_dev=...) advances the counter at the end of the iteration. */
int ar_iter_begin(const int32_t* iter_dev, const int64_t* sched, int batch, int64_t* cur_idx, const float* lr_table, int n_lr,
                  int iters, float* lr_out, ar_stream_t stream);

/* ---- calibration-activation gather ---------------------------------------------------------------------------
 * replaces: torch.cat([inputs[i] for i in indices]) in BlockForwardRunner._select_batch
 *           (auto_round/algorithms/block_runner.py:368-422) and the fp_outputs gather (sign_round/quantizer.py:482).
 * dst[j, :] = src[idx[j], :] for j < n_idx ; rows are row_bytes long (multiple of 16). idx lives on the device. */
int ar_gather_rows(const void* src, const int64_t* idx_dev, void* dst, int64_t n_idx, int64_t row_bytes,
                   ar_stream_t stream);

/* ---- INT packer (GPTQ order) ---------------------------------------------------------------------------------
 * replaces: QuantLinear.pack -> pack_248_bits / pack_3bits of auto_round_extension/torch/qlinear_torch_zp.py:93-263
 *           (zp_off = 1, the "zp-1" convention) and auto_round_extension/torch/qlinear_torch.py:110-281 (zp_off = 0).
 * Wq [out,in] baked fake-quant weight, scale [out, in/gs] (s_dt), zp_tensor [out, in/gs] fp32 or NULL (then
 * zp_scalar).  qweight [in/32*bits, out] int32, qzeros [in/gs, out/32*bits] int32, scales_t [in/gs, out] fp16.
 * Integers are re-derived as rint(Wq/scale + zp) and packed with the reference's exact (additive) arithmetic. */
int ar_pack_int(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar, int64_t out_f,
                int64_t in_f, int gs, int bits, int w_dt, int s_dt, int zp_off, int32_t* qweight, int32_t* qzeros,
                uint16_t* scales_t, ar_stream_t stream);

/* AWQ "GEMM" wire format (4-bit only), the reference's default container for W4 asym under format="auto_round"
 * (export/formats/backends/autoround.py:61-70 -> auto_round:auto_awq).
 * replaces: WQLinear_GEMM.from_linear (auto_round/export/export_to_awq/utils.py:196-274).
 * qweight [in, out/8] int32: eight consecutive OUTPUT channels per word at nibble positions {0,4,1,5,2,6,3,7};
 * qzeros [in/gs, out/8] int32: the zero points packed the same way (stored unchanged); scales [in/gs, out] fp16.
 * Integers are rint(Wq/scale + zp) with the reference's additive (unmasked, wrapping) packing arithmetic. */
int ar_pack_awq(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar, int64_t out_f, int64_t in_f,
                int gs, int w_dt, int s_dt, int32_t* qweight, int32_t* qzeros, uint16_t* scales_t, ar_stream_t stream);

/* ---- MXFP4 / NVFP4 fake-quant -------------------------------------------------------------------------------
 * replaces: quant_mx (auto_round/data_type/mxfp.py:233-291, element rounding :49-85) and
 *           nv_fp4 -> ref_nvfp4_quant -> cast_to_fp4 (auto_round/data_type/nvfp.py:83-98, :67-80, :26-39),
 * for weights (V, max_s tunable) and activations (V = NULL, max_s = NULL).
 * mode 0 = MXFP4 (gs 32; scale_out = shared exponent in the tensor dtype), 1 = NVFP4 (gs 16; scale_out fp32 holding
 * e4m3 values; global_scale_dev = device fp32).  absmax [n_groups] fp32 from ar_group_absmax (weights: computed once). */
int ar_qdq_fp4_fwd(const void* X, const float* V, const float* absmax, const float* max_s, float init_scale,
                   const float* init_scale_dev, const float* global_scale_dev, void* Xq, void* scale_out,
                   int64_t n_groups, int gs, int mode, int x_dt, float lo_bound, float hi_bound, ar_stream_t stream);
/* init_scale_dev (optional, fp32 [n_groups]) replaces the scalar init_scale per group: the searched init scale of the
 * reference's algorithm extension (SignRoundOptimizedWrapperLinear, sign_roundv2/quantizer.py:101-126). */
/* fused backward + sign-SGD for the fp4 weight path (V and max_scale); same contract as ar_qdq_int_bwd_sgd.
 * replaces: torch autograd through quant_mx / nv_fp4 (auto_round/data_type/mxfp.py:233-291, nvfp.py:83-98) followed by
 * SignSGD.step (algorithms/quantization/sign_round/sign_sgd.py:356-389).
 * dV_out/dmax_out (optional) additionally export the raw gradients for parity tests (then no update is applied
 * when lr_v_dev == NULL). */
int ar_qdq_fp4_bwd_sgd(const void* dXq, const void* X, float* V, const float* absmax, float* max_s, float init_scale,
                       const float* init_scale_dev, const float* global_scale_dev, int64_t n_groups, int gs, int mode,
                       int x_dt, float lo_bound,
                       float hi_bound, const float* lr_v_dev, const float* lr_mm_dev, int tune_minmax,
                       const int32_t* snapshot_flag, float* best_V, float* best_max, float* dV_out, float* dmax_out,
                       ar_stream_t stream);

/* ---- init-scale search for the fp4 schemes (algorithm extension) ------------------------------------------------
 * replaces: search_mx_scale (auto_round/data_type/mxfp.py:102-169: candidates 1.0, 0.5, 2.0) and search_nvfp4_scale
 *           (auto_round/data_type/nvfp.py:328-386: 1.0 then 0.50 ... 1.51 step 0.01).
 * For every group: evaluates the fake-quant (V = 0) with max_scale := candidate c for c in candidates (in order) and
 * keeps the first candidate with the strictly smallest importance-weighted squared error
 * sum_k (qdq_k - x_k)^2 * qw_k.  qw_row (optional fp32 [in_pad]) is the per-input-channel importance (imatrix); the
 * weight of element k of group g is qw_row[(g % groups_per_row) * gs + k]; NULL means 1.  best_out fp32 [n_groups]. */
int ar_search_fp4_scale(const void* X, const float* absmax, const float* qw_row, int64_t groups_per_row,
                        const float* global_scale_dev, const float* candidates_dev, int n_candidates, float* best_out,
                        int64_t n_groups, int gs, int mode, int x_dt, ar_stream_t stream);

/* activation fake-quant backward w.r.t. the INPUT (dynamic per-group scale taken from the data itself).
 * replaces: autograd through WrapperLinear._qdq_act (auto_round/wrapper.py:295-321, :530-540) -> quant_mx(x, v=0) /
 * nv_fp4_with_static_gs(x, tensor_max=act_max): the direct path plus the path through the group max (routed to the
 * first index attaining max|x|, times sign(x), like torch.max(dim)).  The forward is ar_qdq_fp4_fwd with V = absmax =
 * max_s = NULL.  dX has the dtype of X. */
int ar_fp4_act_bwd(const void* dXq, const void* X, void* dX, const float* global_scale_dev, int64_t n_groups, int gs,
                   int mode, int x_dt, ar_stream_t stream);

/* ---- FP4 packer ----------------------------------------------------------------------------------------------
 * replaces: qlinear_fp.QuantLinear.pack + _pack_fp4_to_uint8 (auto_round/export/export_to_autoround/qlinear_fp.py
 *           :141-193, :235-265).  packed [out, in/2] uint8 (low nibble = even index), scale_bytes [out, in/gs]
 * (e8m0 for mode 0, e4m3fn for mode 1). */
int ar_pack_fp4(const void* Wq, const void* scale, const float* global_scale_dev, int64_t out_f, int64_t in_f, int gs,
                int mode, int w_dt, uint8_t* packed, uint8_t* scale_bytes, ar_stream_t stream);

/* ---- fused elementwise / normalisation kernels of a Llama-family decoder block (tuning-time block forward/backward) ------
 * replace, for the duration of the tuning loop, what the reference gets from torch.compile(block_forward)
 * (auto_round/utils/device.py:112-122, compressors/base.py:1177-1179) on the module code transformers runs eagerly
 * (transformers/models/llama/modeling_llama.py: LlamaRMSNorm.forward, apply_rotary_pos_emb + repeat_kv, LlamaMLP.forward) and
 * their autograd backwards.  Token-major tensors [rows, features], dt = AR_DT_BF16 | AR_DT_F16.  Forward kernels round where
 * the eager modules round; backward kernels evaluate the exact fp32 derivative and round once.
 *   ar_rmsnorm_fwd   y = w * dt(x * rsqrt(mean(x^2) + eps)); rstd_out [rows] fp32 optional
 *   ar_rmsnorm_bwd   dx = rstd * (dy*w - xhat * mean(dy*w*xhat)) (+ dres: the residual branch's gradient, fused add)
 *   ar_swiglu_fwd    a [rows, F] = dt(silu(g)) * u with g = gu[:, :F], u = gu[:, F:2F], gu row stride ld
 *   ar_swiglu_bwd    in place on gu: (g, u) <- (d g, d u) given da
 *   ar_rope_fwd      qkv [tokens, (hq + 2 hkv) * d] (ld) -> q, k, v [tokens, hq * d]: rotary embedding on q and k (cos / sin
 *                    [batch or 1, seq, d], cs_batch_stride elements between batches, 0 to broadcast), every kv head written
 *                    hq / hkv times (repeat_kv)
 *   ar_rope_bwd      dq, dk, dv [tokens, hq * d] -> dqkv (ld): sums the repeated heads, applies the transposed rotation */
int ar_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd_out, int64_t rows, int hidden, float eps, int dt,
                   ar_stream_t stream);
int ar_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx, int64_t rows,
                   int hidden, int dt, ar_stream_t stream);

/* LayerNorm of OPT / GPT-style blocks (weight and bias, fp32 statistics, one rounding): nn.LayerNorm as transformers'
 * OPTDecoderLayer calls it (modeling_opt.py: self_attn_layer_norm / final_layer_norm) and its autograd backward w.r.t. the input
 * (+ an optional residual gradient); mean / rstd [rows] fp32 are written by the forward and read by the backward. */
int ar_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean_out, float* rstd_out, int64_t rows, int hidden,
                     float eps, int dt, ar_stream_t stream);
int ar_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, const void* dres, void* dx,
                     int64_t rows, int hidden, int dt, ar_stream_t stream);
/* Per-head RMSNorm of the query and key heads of a merged projection output (Qwen3-style q_norm / k_norm:
 * transformers/models/qwen3/modeling_qwen3.py Qwen3Attention.forward, `self.q_norm(self.q_proj(x).view(..., head_dim))`), as the
 * reference reaches it through block_forward.  qkv / out [tokens, ld]: hq query heads, hkv key heads, hkv value heads of d
 * elements (d in {64, 128, 256, 512}); wq / wk [d]; values are copied; rstd_out [tokens, hq + hkv] fp32.  The backward turns the
 * gradient w.r.t. the normalised heads into the gradient w.r.t. the raw projection output in place (value heads untouched). */
int ar_headnorm_fwd(const void* qkv, const void* wq, const void* wk, void* out, float* rstd_out, int64_t tokens, int64_t ld, int hq,
                    int hkv, int d, float eps, int dt, ar_stream_t stream);
int ar_headnorm_bwd(void* dqkv, const void* qkv, const void* wq, const void* wk, const float* rstd, int64_t tokens, int64_t ld, int hq,
                    int hkv, int d, int dt, ar_stream_t stream);
/* dst[c, r] = src[r, c] for 2-byte elements (bf16 / fp16 bit patterns), rows and cols multiples of 64.  The fused block keeps a
 * transposed copy of each fake-quant weight so that the input-gradient GEMM of F.linear (auto_round/wrapper.py:528-556, autograd's
 * dX = dY W) runs with both operands contiguous along the reduction. */
int ar_transpose16(const void* src, void* dst, int64_t rows, int64_t cols, ar_stream_t stream);
int ar_swiglu_fwd(const void* gu, int64_t ld, void* a, int64_t rows, int64_t F, int dt, ar_stream_t stream);
int ar_swiglu_bwd(const void* da, void* gu, int64_t ld, int64_t rows, int64_t F, int dt, ar_stream_t stream);

/* ---- sparse-MoE routing glue: the expert pass over rows sorted by expert -------------------------------------------------
 * replaces: the per-expert `hidden_states[token_idx]`, `* top_k_weights[token_idx, top_k_pos, None]` and
 *           `final_hidden_states.index_add_(0, token_idx, ...)` of MixtralExperts.forward (transformers/models/mixtral/
 *           modeling_mixtral.py) -- the "linear_loop" experts the reference tunes after unfusing them
 *           (auto_round/modeling/fused_moe/moe_experts_interface.py) -- and their autograd mirrors, for rows grouped by expert with
 *           one stable sort: row p of a sorted buffer belongs to token tok[p]; token t's K routed copies are rows pos[t*K + k].
 * ar_moe_expand : out[p, :] = scale[p] * src[tok[p], :]   (scale NULL: plain gather), rows = T*K
 * ar_moe_combine: out[t, :] = res[t, :] + sum_k w[t*K + k] * D[pos[t*K + k], :]   (res / w NULL: 0 / 1); fp32 sum in slot order,
 *                 one rounding -- deterministic, no float atomics
 * ar_moe_rowdot : out[p] = sum_j A[tok[p], j] * B[p, j]  (fp32): the routing-weight gradient
 * H % 8 == 0; tok / pos are int64 device arrays. */
int ar_moe_expand(const void* src, const int64_t* tok, const float* scale, void* out, int64_t rows, int64_t H, int dt,
                  ar_stream_t stream);
int ar_moe_combine(const void* D, const int64_t* pos, const float* w, const void* res, void* out, int64_t T, int64_t H, int K, int dt,
                   ar_stream_t stream);
int ar_moe_rowdot(const void* A, const int64_t* tok, const void* B, float* out, int64_t rows, int64_t H, int dt, ar_stream_t stream);
int ar_rope_fwd(const void* qkv, int64_t ld, const void* cos, const void* sin, int64_t cs_batch_stride, void* q, void* k, void* v,
                int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream);
int ar_rope_bwd(const void* dq, const void* dk, const void* dv, const void* cos, const void* sin, int64_t cs_batch_stride,
                void* dqkv, int64_t ld, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream);

/* ---- the same elementwise work with EAGER TORCH'S OWN ROUNDING POINTS AND SUMMATION ORDER ("exact_rounding") ---------------
 * replaces: the module code the reference's DEFAULT (non-compiled) path runs through block_forward
 *           (auto_round/compressors/utils.py:109-172; transformers/models/llama/modeling_llama.py LlamaRMSNorm.forward,
 *           apply_rotary_pos_emb, LlamaMLP.forward) and torch autograd's backward of it, bit for bit on this GPU: every eager op
 *           rounds to the tensor dtype, row sums follow the association of ATen's reduction kernel (csrc/ar_exact.hip), so a block
 *           run through these kernels follows the reference's own sign-SGD trajectory (auto_round_amd/exact_block.py verifies that
 *           against the module code once per kind of block before using them).  Token-major [rows, features], dt = BF16 | F16.
 *   ar_rmsnorm_fwd_exact  s = dt(x + res) when res != NULL (written to sum_out), else s = x;  y = w * dt(s * rsqrt(mean(s^2) + eps));
 *                         mean = torch's row sum * mean_factor (<= 0: float(rows) / float(rows * hidden)); rstd_out [rows] fp32;
 *                         the rsqrt is evaluated in double and rounded (what torch.rsqrt does on a float tensor in its HIP build);
 *                         flags & 1: the float instruction instead, flags & 2: rstd_out receives the raw row sums (probes)
 *   ar_rmsnorm_bwd_exact  autograd of the above w.r.t. s, op by op (+ dres: the residual branch's gradient, added in dt)
 *   ar_rope_fwd_exact     q [tokens, hq*d] (row stride ldq), k [tokens, hkv*d] (ldk) -> rotated, contiguous; cos / sin as ar_rope_fwd
 *   ar_rope_bwd_exact     gradients of the rotated q / k, [B, S, h, d] through element strides (batch, token, head) -> d q, d k
 *                         (token-major, row strides lddq / lddk: column slices of one merged gradient buffer are fine)
 *   ar_swiglu_fwd_exact   a = dt(dt(silu(g)) * u), g / u separate [rows, F] matrices with row strides
 *   ar_swiglu_bwd_exact   dg, du [rows, F] (row strides lddg / lddu) from da, g, u; contract != 0: `1 + g * (1 - s)` as the fma ATen's silu_backward kernel has
 * Unsupported shapes (rows < 8, hidden < 256, ...: another ATen code path) return AR_ERR_UNSUPPORTED and the caller keeps torch. */
int ar_rmsnorm_fwd_exact(const void* x, const void* res, const void* w, void* sum_out, void* y, float* rstd_out, int64_t rows, int hidden,
                         float eps, float mean_factor, int flags, int dt, ar_stream_t stream);
int ar_rmsnorm_bwd_exact(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx, int64_t rows,
                         int hidden, int dt, ar_stream_t stream);
int ar_rope_fwd_exact(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* cos, const void* sin, int64_t cs_batch_stride,
                      void* q_out, void* k_out, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream);
int ar_rope_bwd_exact(const void* gq, int64_t q_sb, int64_t q_ss, int64_t q_sh, const void* gk, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                      const void* cos, const void* sin, int64_t cs_batch_stride, void* dq, int64_t lddq, void* dk, int64_t lddk, int64_t tokens,
                      int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream);
int ar_swiglu_fwd_exact(const void* g, int64_t ldg, const void* u, int64_t ldu, void* a, int64_t rows, int64_t F, int dt,
                        ar_stream_t stream);
int ar_swiglu_bwd_exact(const void* da, const void* g, int64_t ldg, const void* u, int64_t ldu, void* dg, int64_t lddg, void* du, int64_t lddu,
                        int64_t rows, int64_t F, int contract, int dt, ar_stream_t stream);

/* ---- nn.LayerNorm with the bits of torch's own kernels (round 6; csrc/ar_exact_ln.hip) -------------------------------------------
 * replaces: the two LayerNorms of an OPT-style decoder block as the reference's DEFAULT (eager) path runs them through block_forward
 *           (auto_round/compressors/utils.py:109-172; transformers/models/opt/modeling_opt.py OPTDecoderLayer.self_attn_layer_norm /
 *           final_layer_norm): under autocast `layer_norm` is an fp32 op -- x.float() -> at::native::vectorized_layer_norm_kernel
 *           <float, float, false> -> the next linear's cast to the activation dtype -- and torch autograd's backward of it
 *           (at::native::layer_norm_grad_input_kernel<float, float, false>); both kernels restated from the gfx950 code objects of
 *           the installed torch, the two dtype conversions folded in (auto_round_amd/exact_opt_block.py proves the result against
 *           the module code before it is used).  x / gamma / beta / y / dy / dx in dt (BF16 | F16), mean / rstd fp32 [rows].
 *   ar_layernorm_fwd_exact   y = dt(fma(rstd * (x - mean), gamma, beta)) with ATen's Welford statistics (per-thread online update,
 *                            shuffle-down and shared-memory combines); mean_out / rstd_out may be NULL; flags bit 0: IEEE 1/x
 *                            instead of v_rcp_f32 in the Welford updates, bit 1: rsqrt evaluated in double (other torch builds)
 *   ar_layernorm_bwd_exact   dx = dt(ATen's grad-input formula) (+ dres, added in dt: the residual branch's gradient); rows < 32768
 *                            (above that ATen's ROCm build runs another kernel)
 * hidden % 4 != 0, unaligned pointers, rows >= 32768 (backward): AR_ERR_UNSUPPORTED and the caller keeps torch's own ops. */
int ar_layernorm_fwd_exact(const void* x, const void* gamma, const void* beta, void* y, float* mean_out, float* rstd_out, int64_t rows,
                           int hidden, float eps, int flags, int dt, ar_stream_t stream);
int ar_layernorm_bwd_exact(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd, const void* dres, void* dx,
                           int64_t rows, int hidden, int dt, ar_stream_t stream);

/* ---- weight-gradient GEMM (hand-written MFMA, gfx950) -------------------------------------------------------------
 * replaces: the autograd backward of F.linear(x, weight_q) with respect to weight_q inside WrapperLinear.forward
 *           (auto_round/wrapper.py:528-556): dW[M,N] = dY^T X, dY [K,M] and X [K,N] row-major bf16 (leading dimensions in
 *           elements), fp32 accumulation, one rounding to bf16; accumulate != 0 adds the previous bf16 dW before rounding
 *           (torch addmm_, the gradient-accumulation case).  M, N multiples of 256, K >= 96 (a K that is not a multiple of 128 is completed with zero rows); operands
 *           16-byte aligned, ldy/ldx multiples of 8, ldw a multiple of 4; anything else returns AR_ERR_UNSUPPORTED and the
 *           caller keeps the library GEMM.  Needs 128 KB of dynamic LDS per workgroup. */
int ar_gemm_dw(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
               int64_t ldw, int accumulate, void* workspace, int64_t workspace_bytes, ar_stream_t stream);
/* The same GEMM with the summation structure chosen by the caller: nsplit = 1 is one pass over K in token order, nsplit >= 2 splits
 * K into that many contiguous slices of whole 128-row chunks whose fp32 partial tiles (workspace: nsplit * M * N * 4 bytes) are
 * summed in slice order.  exact_rounding (auto_round_amd/exact_block.py) uses it to reproduce, bit for bit, the weight gradient
 * the library GEMM behind torch autograd's `grad_output.t().mm(input)` (auto_round/wrapper.py:528-556) returns for a shape --
 * whichever structure proves equal on the installed stack.  A structure that cannot be delivered returns AR_ERR_UNSUPPORTED. */
int ar_gemm_dw_ex(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
                  int64_t ldw, int accumulate, void* workspace, int64_t workspace_bytes, int nsplit, ar_stream_t stream);
/* The same GEMM with a stream-K summation structure given by the caller: tile t (row-major id of the 256 x 256 output tiles) is
 * summed in one pass over K when kcut[t] == 0, else in two parts, k-rows [0, kcut[t]) and [kcut[t], K) (a multiple of 32), each
 * from a zero accumulator, added in fp32, rounded once (an entry that is not such a multiple of 32 inside (0, K) means one pass).
 * Reproduces the library kernel behind the same autograd GEMM
 * (auto_round/wrapper.py:528-556) when that kernel streams its last tiles over a fixed workgroup grid (auto_round_amd/streamk.py
 * finds the structure).  kcut is a device array of tiles entries; workspace = tiles * 256 * 256 * 4 bytes, 16-byte aligned. */
int ar_gemm_dw_sk(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
                  int64_t ldw, void* workspace, int64_t workspace_bytes, const int32_t* kcut, ar_stream_t stream);
/* caller-owned scratch ar_gemm_dw wants for (M, N, K) (the library never allocates): 0 when the output tiles alone fill the
 * chip; otherwise the fp32 partial tiles of its split-K form (few tiles, deep K -- e.g. OPT-125M's 768x768 weight against 16384
 * tokens), which are summed in slice order, i.e. deterministically.  Without the workspace the call still works, unsplit. */
int64_t ar_gemm_dw_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* The same weight-gradient GEMM GROUPED over the experts of a sparse-MoE block: group e owns k-rows [row_off[e], row_off[e+1]) of
 * dY [R, M] and X [R, N] (the rows of the tokens routed to expert e, sorted by expert) and writes its own dW_e [M, N] = dY_e^T X_e
 * to dW + w_off[e] (elements, ldw).  ONE launch of n_groups * (M/256) * (N/256) workgroups; row_off ([n_groups + 1] int32) and
 * w_off ([n_groups] int64, multiples of 8 elements) are DEVICE arrays, so the launch needs no host knowledge of the row counts (hipGraph-capturable); a group
 * without rows writes zeros (sign(0) = 0: the sign-SGD step then leaves its parameters untouched, as a missing gradient does).
 * replaces: the autograd backward of the per-expert F.linear calls in the reference's "linear loop" experts
 *           (auto_round/modeling/fused_moe/moe_experts_interface.py:173-289 around auto_round/wrapper.py:528-556).  Deterministic. */
int ar_gemm_dw_grouped(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t ldy, int64_t ldx, int64_t ldw,
                       const int32_t* row_off, const int64_t* w_off, int n_groups, ar_stream_t stream);

/* ---- forward / input-gradient GEMM (hand-written MFMA "NT" kernel, gfx950; SURVEY 8 row f1 forward side) ---------------------
 * replaces: the forward of F.linear(x, weight_q) inside WrapperLinear.forward (auto_round/wrapper.py:528-556): C[M,N] = A[M,K] B[N,K]^T,
 *           A = activations [tokens, in], B = the fake-quant weight [out, in], both K-contiguous bf16 (leading dimensions in elements),
 *           fp32 accumulation over K in ascending order (default: v_mfma_f32_16x16x32_bf16; ar_gemm_nt_config(0): 32x32x16 -- identical
 *           bits in both shapes, an EMPIRICAL result held by tests/test_gpu_gemm_nt.py), one rounding to bf16 -- the
 *           summation the library's kernel performs for these shapes; with B = a transposed weight copy [in, out] the same call is the
 *           input gradient dX = dY Wq.  M any (rows past M are clamped on the load side, masked on the store side), N % 256 == 0,
 *           K % 128 == 0, operands 16-byte aligned, lda / ldb multiples of 8, ldc a multiple of 4; anything else returns
 *           AR_ERR_UNSUPPORTED and the caller keeps the library GEMM.  No bias.  Needs 128 KB of dynamic LDS per workgroup. */
int ar_gemm_nt(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               ar_stream_t stream);
/* Grouped over the experts of a sparse-MoE block (one launch per projection instead of a Python loop of per-expert GEMMs with a host
 * read of the token counts, auto_round/modeling/fused_moe/moe_experts_interface.py:173-289): group e multiplies rows
 * [row_off[e], row_off[e+1]) of A (and writes the same rows of C) with ITS matrix B + b_off[e] ([N, K], ldb).  row_off
 * ([n_groups + 1] int32) and b_off ([n_groups] int64, elements, EVERY ENTRY A MULTIPLE OF 8: 16-byte LDS-DMA reads; device values
 * cannot be validated here -- auto_round_amd/fused_block.py checks its arena offsets) are DEVICE arrays; M = row_off[n_groups] = rows of A.  The grid covers
 * M / 256 + n_groups row tiles (every group may end in a partial tile); workgroups past the last real tile exit.  Deterministic. */
int ar_gemm_nt_grouped(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                       const int32_t* row_off, const int64_t* b_off, int n_groups, ar_stream_t stream);
/* experiment knob of the two entry points above for tools/gpu/r05_gemm_nt*_probe.py (binding hygiene): variant 3 = the kernel on
 * v_mfma_f32_16x16x32_bf16 (default since round 5), 0 = on 32x32x16 with the LDS-DMA pieces issued between the MFMAs, 1 = 32x32x16 with
 * the pieces at the end of the fragment-read part of every second phase, 2 = 32x32x16 with one wave per SIMD and 128 x 128 wave tiles;
 * all four produce identical bits; -1 keeps.  Returns the variant in use. */
int ar_gemm_nt_config(int variant);
/* measurement hygiene (tools/gpu/r05_gemm_nt_trace.py; no reference counterpart): ar_gemm_nt with s_memtime bookkeeping -- trace receives
 * [row tiles * column tiles][8 waves][8] uint64: shader cycles a wave spent, summed over its phases, in the fragment-read part, parked at
 * the barrier after it, in the MFMA part, parked at the barrier after it; the cycles of the whole K loop; the number of phases; the whole
 * K loop in ticks of the constant 100 MHz counter (s_memrealtime: cycles / ticks * 100 MHz = the shader clock the kernel ran at); 0. */
int ar_gemm_nt_trace(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     unsigned long long* trace, int variant, ar_stream_t stream);
/* experiment knobs of the kernel above for tools/gemm_dw_probe.py (binding hygiene; -1 keeps a value): sem = lane->piece rule
 * of the transposing LDS read (1 | 2), order = tile order (0 identity, 1 XCD chunks, 2 XCD 2x8 patches); sem 20 | 21 = hybrid split of the
 * last partial round off | on; sem 32 | 30 | 31 = the kernel on v_mfma_f32_16x16x32_bf16 (default since round 5) | on 32x32x16 (rounds
 * 2-4) | on 32x32x16 with the LDS-DMA issued outside the MFMA cluster -- all three produce identical bits.  Returns sem*10+order. */
int ar_gemm_dw_config(int sem, int order);

/* ---- causal attention forward (hand-written MFMA flash attention, gfx950) ------------------------------------------
 * replaces: the attention forward of the decoder block inside the tuning loop -- transformers' sdpa_attention_forward
 *           (transformers/integrations/sdpa_attention.py: torch scaled_dot_product_attention(q, k, v, is_causal=True)), which the
 *           reference reaches through block_forward (auto_round/utils/model.py block_forward, compressors/base.py:1177-1179).
 *           Q, K, V, O: [B, S, H, D] token-major bf16 (K / V already repeated to H heads), D = 128 or 64, S a multiple of 128;
 *           LSE [B, H, S] fp32 = natural-log row sums of the scaled scores, the form
 *           aten::_scaled_dot_product_efficient_attention_backward consumes.  scale = 1/sqrt(D) for the stock models.
 *           ldq / ldkv: elements between consecutive tokens of Q and of K / V (0 = H * D, i.e. contiguous [B, S, H, D]); larger
 *           strides let the operands be column slices of ONE merged q/k/v projection output (OPT: [tokens, 3 H D]) -- no copies;
 *           `scale` then also carries a q scaling the module applies before the attention (OPTAttention: q_proj(x) * head_dim^-0.5,
 *           exact to fold when it is a power of two).
 *           Anything else (no causal mask, other head sizes) returns AR_ERR_UNSUPPORTED and the caller keeps torch's SDPA. */
int ar_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                float scale, int causal, int64_t ldq, int64_t ldkv, ar_stream_t stream);
/* The same forward for the attention mask the reference's calibration flow really hands to a block
 * (auto_round/calibration/llm.py:360-402: attention_mask = ones with the last position cleared -> transformers >= 5 builds the boolean
 * [1, 1, S, S] mask `causal & key-is-valid`; auto_round/calibration/inputs.py:100-107 casts it to the amp dtype, i.e. a 0 / 1 ADDITIVE
 * bias): softmax(scale * Q K^T + bias), bias(q, k) = bias_in where k <= q and k < valid_len, bias_out elsewhere -- both finite, so
 * every query attends to every key (no tile is skipped) and the mask costs two registers instead of an [S, S] operand.  LSE includes
 * the bias (what aten::_scaled_dot_product_efficient_attention_backward with the same attn_bias expects).  Hard masks (-inf) and
 * unstructured biases return AR_ERR_UNSUPPORTED: the caller keeps torch's SDPA. */
int ar_attn_fwd_masked(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                       float scale, float bias_in, float bias_out, int64_t valid_len, int64_t ldq, int64_t ldkv, ar_stream_t stream);

/* ---- attention forward with the library's bits (csrc/ar_attn_exact.hip; round 6) ---------------------------------------------
 * replaces: the same call as ar_attn_fwd_masked -- transformers' sdpa_attention_forward -> F.scaled_dot_product_attention(q, k, v,
 *           attn_mask = the calibration flow's 0 / 1 additive mask) under the reference's block_forward
 *           (auto_round/compressors/utils.py:109-172, auto_round/calibration/llm.py:360-402, inputs.py:100-107) -- on the
 *           BIT-IDENTICAL paths: O and LSE equal, value for value, what torch 2.10.0+rocm7.0 returns for that call on gfx950 (AOTriton
 *           0.11.1 `attn_fwd` with an additive bias; the arithmetic was read off the shipped code objects: key blocks of 32 keys at head
 *           size 64 and 64 at head size 128, the bias scaled in bf16, the block row sums in the bias tile's lane order, ...: the
 *           kernel's header).  Equality is an empirical property of one library build: callers prove it per call signature against
 *           torch before using it (auto_round_amd/exact_block.py `plan_against_module`).
 *           Q / K / V bf16 with element strides (batch, head, token; unit stride along D): any of [B, S, H, D] / [B, H, S, D] views;
 *           K / V hold H / kv_rep heads (query head h reads key head h / kv_rep: transformers' repeat_kv only copies).  O is written
 *           token-major contiguous [B, S, H, D]; LSE [B, H, S] fp32 (natural log).  bias_in / bias_out / valid_len: ar_attn_fwd_masked's
 *           structured mask; both values must be bf16 numbers.  key_block: the keys per online-softmax step -- the one tile size of the
 *           library's configuration the bits depend on (it picks the configuration by sequence length and head size): 0 = the tuning
 *           minibatch's (64 at head size 128, 32 at 64; what the library uses for S = 1024 .. 2048, at head size 128 also 4096), or 16 /
 *           32 / 64 (callers try them in their proof: auto_round_amd/exact_block.py).  D in {64, 128}, S % 128 == 0; anything else
 *           AR_ERR_UNSUPPORTED. */
int ar_attn_fwd_exact(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                      int64_t kv_rep, float scale, float bias_in, float bias_out, int64_t valid_len, int64_t q_bs, int64_t q_hs,
                      int64_t q_ts, int64_t k_bs, int64_t k_hs, int64_t k_ts, int64_t v_bs, int64_t v_hs, int64_t v_ts,
                      int64_t key_block, ar_stream_t stream);

/* The matching BACKWARD with the library's bits: what aten::_scaled_dot_product_efficient_attention_backward returns for the same call
 * on this stack (AOTriton 0.11.1 bwd_preprocess + bwd_kernel_dk_dv + bwd_kernel_dq with an additive bias; autograd of the call
 * ar_attn_fwd_exact replaces; the arithmetic -- delta in the library's lane order, scores accumulated on top of fl(bias / sm_scale),
 * p = exp2(fma(..)) with the one non-fused element of the key-side kernel, natural / adopted k orders of the accumulating products -- is
 * in csrc/ar_attn_exact.hip).  Q / K / V / O / dO bf16 with element strides (batch, head, token), K / V grouped as in the forward;
 * O / LSE are the forward's results.  dQ, dK, dV are written token-major [B, S, H, D] with token strides lddq / lddk / lddv (0 = H * D;
 * larger strides: column slices of a merged buffer); dK / dV hold one gradient per QUERY head -- the caller adds the kv_rep heads of a
 * group, as autograd's expand backward does.  workspace: ar_attn_bwd_exact_workspace_bytes(B, S, H) bytes.  D in {64, 128},
 * S % 128 == 0, S <= 4096; anything else AR_ERR_UNSUPPORTED.  Equality with the library is proven per call signature by the caller. */
int64_t ar_attn_bwd_exact_workspace_bytes(int64_t B, int64_t S, int64_t H);
int ar_attn_bwd_exact(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                      void* dV, int64_t B, int64_t S, int64_t H, int64_t D, int64_t kv_rep, float scale, float bias_in, float bias_out,
                      int64_t valid_len, int64_t q_bs, int64_t q_hs, int64_t q_ts, int64_t k_bs, int64_t k_hs, int64_t k_ts, int64_t v_bs,
                      int64_t v_hs, int64_t v_ts, int64_t o_bs, int64_t o_hs, int64_t o_ts, int64_t do_bs, int64_t do_hs, int64_t do_ts,
                      int64_t lddq, int64_t lddk, int64_t lddv, void* workspace, int64_t workspace_bytes, ar_stream_t stream);

/* Launch form of ar_attn_fwd_exact / ar_attn_bwd_exact (binding hygiene: a tuning knob for A/B measurements, results are identical
 * in every form -- the per-row arithmetic does not depend on how many rows a workgroup owns).  Bits 0-1 forward, bits 2-3 backward:
 * 0 default, 1 workgroups of 4 waves (128 own rows), 2 workgroups of 8 waves (256), 3 (backward, head size 128) dK and dV in one
 * kernel on one wave per SIMD -- the default there.  cfg < 0 only reads.  -> the previous value. */
int ar_attn_exact_config(int cfg);

/* ---- causal attention backward, head size 64 (deterministic: two MFMA kernels, no float atomics) ---------------------------
 * replaces: autograd of the same attention call -- torch's aten::_scaled_dot_product_efficient_attention_backward, i.e. aiter's
 *           fmha_bwd (+ pre / post-process kernels), which accumulates dQ with fp32 atomics (0.42 ms per call at OPT-125M's minibatch,
 *           the largest kernel of BASELINE configs[0]).  Q, K, V, dO and the outputs dQ, dK, dV are token-major [B, S, H, 64] bf16 with
 *           caller-given row strides (ld*: elements between consecutive tokens, 0 = H * 64): inputs and outputs may be column slices
 *           of merged [tokens, 3 H 64] buffers, so the gradient of a merged q/k/v projection needs no gather pass.  O [B, S, H, 64]
 *           and LSE [B, H, S] are ar_attn_fwd's results.  workspace: ar_attn_bwd_workspace_bytes(B, S, H) bytes of scratch
 *           (D = rowsum(dO * O) and lse * log2(e)).  S % 256 == 0, S <= 4096, causal only; anything else AR_ERR_UNSUPPORTED (the
 *           caller keeps the library backward; the CAUSAL head size 128 is left to it on purpose: the library's asm kernel is faster). */
int64_t ar_attn_bwd_workspace_bytes(int64_t B, int64_t S, int64_t H);
int ar_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK, void* dV,
                int64_t B, int64_t S, int64_t H, int64_t D, float scale, int causal, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, void* workspace, int64_t workspace_bytes, ar_stream_t stream);
/* The same backward (head size 64 or 128, deterministic) for the calibration flow's structured additive mask (ar_attn_fwd_masked:
 * bias_in where k <= q and k < valid_len, bias_out elsewhere; auto_round/calibration/llm.py:360-402, inputs.py:100-107): every
 * (query, key) pair contributes, no tile is skipped.  O / LSE are ar_attn_fwd_masked's results.  Head size 128 runs the key side as two
 * kernels (dV, dK) and then the query side: 8 GEMM passes, 3.0 ms where torch's additive-bias backward takes 5.4 at 8 x 32 x 2048 x 128. */
int ar_attn_bwd_masked(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                       void* dV, int64_t B, int64_t S, int64_t H, int64_t D, float scale, float bias_in, float bias_out, int64_t valid_len,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv,
                       void* workspace, int64_t workspace_bytes, ar_stream_t stream);

/* ---- optional device-side timing of the hot kernels (bench.py / tools; OFF by default) --------------------------
 * binding hygiene / measurement, no reference counterpart (the reference times blocks on the host,
 * compressors/orchestrator.py:792-794).  While enabled, each profiled launch carries a start/stop event pair on the dispatch
 * itself (hipExtLaunchKernelGGL): ar_profile_read() sums the kernels' own durations -- the figure rocprofv3 --kernel-trace
 * reports -- over the recorded launches of `kernel_id` that processed at least `min_units` units (groups for the quant
 * kernels, output elements for the GEMM), so that bench.py's live roofline fraction is honest for an 11 us kernel too.
 * These three calls are the only ones that allocate (events) or synchronise (read / reset). */
enum { AR_PROF_INT_FWD = 0, AR_PROF_INT_BWD = 1, AR_PROF_FP4_FWD = 2, AR_PROF_FP4_BWD = 3, AR_PROF_GEMM_DW = 4,
       AR_PROF_NORM = 5, AR_PROF_SWIGLU = 6, AR_PROF_ROPE = 7, AR_PROF_GEMM_NT = 8 };
int ar_profile_enable(int on);
int ar_profile_reset(void);
int ar_profile_read(int kernel_id, int64_t min_units, double* total_ms, double* min_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* AR_MI355X_H */
