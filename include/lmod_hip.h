/* lmod_hip.h — C ABI of the MI355X (gfx950) kernels behind the LLaVA-MoD distillation step.
 *
 * The reference (shufangxun/LLaVA-MoD) is pure Python and has NO native interface of its own:
 * every op below is what the reference reaches through torch / HF transformers / DeepSpeed at the
 * cited call site.  This header is therefore the boundary a maintainer binds with ctypes
 * (see INTEGRATION.md); nothing in it carries a torch type.
 *
 * Conventions: raw device pointers, explicit dims / leading dimensions (in ELEMENTS), the HIP
 * stream to enqueue on, caller-owned outputs and workspaces.  Every function only enqueues work
 * (no allocation, no synchronisation) and returns 0 on success or a negative LMOD_E* code; it
 * never throws.  bf16 tensors are `void*` to raw uint16 bit patterns.
 *
 * State.  No entry point keeps data between calls: every buffer, workspace and counter is the caller's.  What the library
 * does keep, process-wide, is launch CONFIGURATION, all of it idempotent and safe under concurrent first use:
 *   - per kernel instance, a flag "dynamic-LDS attribute already set" (hipFuncSetAttribute is issued once);
 *   - the compute-unit count per device id (the persistent GEMM grid is one workgroup per CU of the CURRENT device);
 *   - the communicator handles behind lmod_comm_* (owned by the caller through their opaque pointers);
 *   - measurement switches from the environment, documented beside their readers in csrc/ — read once at first use:
 *     LMOD_GEMM_WAVES, LMOD_ATTN_FWD, LMOD_ATTN_BWD, LMOD_ATTN_BWD64, LMOD_ATTN_BWD_SPLIT, LMOD_ATTN_XCD, LMOD_WGRAD_SPLIT; read at every launch (so that one test process can
 *     run both arms of an A/B): LMOD_GEMM_PERSIST, LMOD_GEMM_PERSIST_GROUPED, LMOD_GEMM_PERSIST_ROUNDS, LMOD_GEMM_KV4, LMOD_GEMM_SB4, LMOD_GEMM_TN4, LMOD_GEMM_TILE.
 *     Unset, they select the shipped routing.  The GEMM switches' arms are bit-identical; the attention switches choose between kernels
 *     that pass the same tests against the fp32 reference but round differently (tests/test_kernels_gpu.py).
 */
#ifndef LMOD_HIP_H
#define LMOD_HIP_H
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMOD_OK 0
#define LMOD_EINVAL (-1)       /* bad pointer / shape / alignment */
#define LMOD_ELAUNCH (-2)      /* HIP reported a launch error */
#define LMOD_EUNSUPPORTED (-3) /* valid request outside the compiled envelope */

/* ---- GEMM ------------------------------------------------------------------------------------
 * C[b] (+)= act(A[b] (M x K, row-major, lda) * B[b]^T (N x K, row-major, ldb) + bias[N]).
 * Replaces nn.Linear forward: qwen2/modeling_qwen2.py:262-264,320 (QKV/O), :186-187 (SwiGLU MLP),
 * :1163 (lm_head); multimodal_projector/builder.py:57-61; HF CLIP linears + patch conv
 * (multimodal_encoder/clip_encoder.py:54).  With m_valid/k_valid (device int[batch]) it is the
 * grouped GEMM over DeepSpeed MoE capacity slabs (deepspeed.moe.experts; llava_qwen2_moe.py:536-546).
 * act: 0 none, 1 exact GELU, 2 quick_gelu.  out_f32: C is fp32.  accumulate: C += .
 * Requires K % 8 == 0, lda/ldb % 8 == 0, A/B 16-byte aligned. */
int lmod_gemm_bf16_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int lda, int ldb,
                      int ldc, int batch, long long strideA, long long strideB, long long strideC, const int* m_valid,
                      const int* k_valid, int act, int out_f32, int accumulate, hipStream_t stream);

/* C[M x N] = bf16(res + bf16(A W^T)): a bias-free projection whose output joins the residual stream — the decoder layer's
 * hidden_states = residual + self_attn(...) / residual + mlp(...) (qwen2/modeling_qwen2.py:757-775; MoE layer:
 * llava_qwen2_moe.py:143-167) with the add in the GEMM epilogue.  Same roundings as lmod_gemm_bf16_nt followed by the residual
 * path of lmod_rmsnorm_fwd (bit-identical); the norm that follows then reads one tensor and writes one.  Only shapes that
 * lmod_gemm_bf16_nt runs on the 4-wave 256-tile kernel (M >= 512, N >= 256, enough tiles): others return LMOD_EUNSUPPORTED and the
 * caller keeps the two-step form.  N % 8 == 0, ldc / ldr % 8 == 0, 16-byte aligned pointers, C != res, bias must be NULL. */
int lmod_gemm_bf16_nt_res(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int lda,
                          int ldw, int ldc, int ldr, hipStream_t stream);

/* Fused q/k/v projection + rotary embedding: C[M x N] = rope(A W^T + bias), W = [Wq; Wk; Wv] stored as ONE [N x K] matrix.
 * Replaces `self.q_proj / k_proj / v_proj` followed by `apply_rotary_pos_emb` (qwen2/modeling_qwen2.py:262-264, :146-171;
 * tables :119-134) in ONE launch: heads are 128 wide, columns [0, rope_cols) hold the q and k heads and are rotated in the GEMM
 * epilogue with the bf16 cos/sin rows [max_pos x 128] of position pos[row] (int32 [M]); columns >= rope_cols (v) are stored
 * as computed.  Results are bit-identical to lmod_gemm_bf16_nt + lmod_rope.
 * Requires K % 8 == 0, N % 16 == 0, rope_cols % 256 == 0, lda/ldw/ldc % 8 == 0, A/W/C 16-byte aligned. */
int lmod_gemm_qkv_rope_bf16(const void* A, const void* W, void* C, const void* bias, int M, int N, int K, int lda, int ldw,
                            int ldc, const void* cos_t, const void* sin_t, const int* pos, int rope_cols, hipStream_t stream);

/* Fused SwiGLU MLP input half: act_out[b] (M x N) = silu(A Wg^T) * (A Wu^T), W = [Wg; Wu] stored as ONE [2N x K]
 * matrix (gate rows first — the fused gate_proj/up_proj weight; qwen2/modeling_qwen2.py:175-187, MoE experts
 * llava_qwen2_moe.py:536-546).  gu_out (may be NULL): [M x 2N] bf16 pre-activations [gate | up] kept for backward.
 * m_valid as in lmod_gemm_bf16_nt; rows m_valid..roundup8(m_valid)-1 of act_out are written as zeros.
 * Requires K % 8 == 0, N % 8 == 0, 16-byte aligned pointers, leading dims % 8 == 0. */
int lmod_gemm_swiglu_bf16(const void* A, const void* W, void* act_out, void* gu_out, int M, int N, int K, int lda,
                          int ldw, int ld_act, int ld_gu, int batch, long long strideA, long long strideW,
                          long long stride_act, long long stride_gu, const int* m_valid, hipStream_t stream);

/* Down-projection dgrad with the SwiGLU backward in its epilogue: dact = A (M x K) * Bt^T (Bt = W_down^T, N x K), then
 * dgu = [dact * up * silu'(gate) | dact * silu(gate)] (M x 2N) from the saved pre-activations gu = [gate | up];
 * d(act) is never written (autograd of `down_proj(act_fn(gate_proj(x)) * up_proj(x))`, qwen2/modeling_qwen2.py:186-187).
 * dgu may alias gu.  Grouped use as lmod_gemm_bf16_nt; rows m_valid..roundup8(m_valid)-1 of dgu are zeroed.
 * Requires K % 8 == 0, N % 16 == 0, 16-byte aligned pointers, leading dims % 8 == 0. */
int lmod_gemm_swiglu_bwd_bf16(const void* A, const void* Bt, const void* gu, void* dgu, int M, int N, int K, int lda,
                              int ldb, int ld_gu, int ld_dgu, int batch, long long strideA, long long strideB,
                              long long stride_gu, long long stride_dgu, const int* m_valid, hipStream_t stream);

/* Weight-gradient accumulate: C (fp32, M x N) += At (M x K) * X, K = tokens (main_grad accumulation of nn.Linear
 * weights, dW = dY^T X; the reference gets this from autograd + DeepSpeed's fp32 gradient accumulation).
 * At = dY^T [M x K] (K-contiguous).  X: b_kmajor 0 -> Bt [N x K] (K-contiguous); b_kmajor 1 -> the layer input as
 * autograd holds it, [K x N] with row stride ldb (read through transposing LDS reads, no transposed copy).
 * b_kmajor 2 -> BOTH operands as autograd holds them: At is dY [K x M] (row stride lda), B is X [K x N]; no transposed
 * copy of either (the step's path: grad_output.t() @ input of nn.Linear's backward, reference linears as in
 * lmod_gemm_bf16_nt).  Mode 2 requires M % 8 == 0, N % 8 == 0, lda / ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned
 * pointers; any K; operand windows past 2 GiB are walked in K chunks.
 * Few-tile outputs use deterministic split-K.  workspace: 16-byte aligned device memory, zeroed ONCE by the caller,
 * used by one stream at a time: 16 KiB of tile semaphores (self-resetting) + up to 8 partial images of 256 KiB per
 * 256x256 tile; its size bounds the split (NULL: no split).  The last split to arrive adds all partials in split
 * order, so the result does not depend on scheduling. */
int lmod_gemm_wgrad_bf16_nt(const void* At, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int b_kmajor, void* workspace, long long workspace_bytes, hipStream_t stream);

/* Weight-gradient GEMM: C[b] (M x N) (+)= A[b]^T * B[b], A [K x M] (lda), B [K x N] (ldb) — dW = dY^T X on the
 * token-major tensors autograd holds (the implicit grad_output.t() @ input of nn.Linear's backward; reference
 * linears as in lmod_gemm_bf16_nt).  k_valid (device int[batch]): live reduction rows per batch (MoE capacity slots).
 * Split-K: batch = S, strideA = Kc*lda, strideB = Kc*ldb, C = [S, M, N] workspace.
 * Requires M % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned pointers. */
int lmod_gemm_bf16_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int batch,
                      long long strideA, long long strideB, long long strideC, const int* k_valid, int out_f32,
                      int accumulate, hipStream_t stream);

/* out[C x ld_out] = in[R x C]^T, zero-filling columns R..ld_out-1 (makes dgrad / wgrad operands
 * K-contiguous for lmod_gemm_bf16_nt; autograd's implicit .t() in the reference).  r_valid (nullable, [batch]): live rows per batch entry (MoE capacity
 * slabs): rows past it are not transposed at all — the k_valid GEMM that consumes the result never reads them — except that
 * the 8-column group holding the boundary is zero-filled. */
int lmod_transpose_bf16(const void* in, void* out, int R, int C, int ld_in, int ld_out, int batch,
                        long long stride_in, long long stride_out, const int* r_valid, hipStream_t stream);

/* ---- row kernels -----------------------------------------------------------------------------
 * Qwen2RMSNorm (qwen2/modeling_qwen2.py:83-97) with the decoder layer's residual add fused
 * (:757-775, llava_qwen2_moe.py:143-167): h = res ? bf16(x+res) : x; y = w * bf16(h*rstd).
 * res, h_out, rstd may be NULL. */
int lmod_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd, int T, int H,
                     float eps, hipStream_t stream);
/* dh = rmsnorm_bwd(dy; h, w, rstd) + dres (dres may be NULL).  Norm weights are frozen in every
 * stage of the reference (llava_qwen2_moe.py:501-506), so no dw. */
int lmod_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dh,
                     int T, int H, hipStream_t stream);
/* nn.LayerNorm forward of the frozen CLIP tower (clip_encoder.py:45 runs under no_grad). */
/* Weight gradients of the row parameters (only when they are trainable; the distillation shells freeze them):
 * lmod_rmsnorm_dw: dw[H] (+)= sum_t dy[t,:] * bf16(h[t,:] * rstd[t])  (Qwen2RMSNorm.weight; deterministic; workspace 64*H floats);
 * lmod_embed_wgrad: dW[idx[r], :] += d_embeds[r, :] for idx[r] >= 0  (embed_tokens.weight through the splice map; fp32 atomics). */
int lmod_rmsnorm_dw(const void* dy, const void* h, const float* rstd, float* dw, float* workspace, int T, int H,
                    int accumulate, hipStream_t stream);
int lmod_embed_wgrad(const void* d_embeds, const int* idx, float* dW, long long rows, int H, hipStream_t stream);
int lmod_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int T, int H, float eps,
                       hipStream_t stream);
/* apply_rotary_pos_emb (qwen2/modeling_qwen2.py:146-171), in place on the first `nheads` heads
 * of each row of buf[T x ld]; cos/sin: [max_pos x hd] bf16; pos: int32[T]; backward != 0 applies
 * the gradient map. */
int lmod_rope(void* buf, const void* cos_t, const void* sin_t, const int* pos, int T, int nheads, int hd, int ld,
              int backward, hipStream_t stream);
/* Qwen2MLP activation (qwen2/modeling_qwen2.py:186-187): out = bf16(silu(gate)) * up. */
/* seg_valid (device int[rows/seg_rows]) or NULL: rows r with r % seg_rows >= seg_valid[r / seg_rows] are not
 * read and are written as zeros (dead MoE capacity slots stay finite). */
int lmod_swiglu_fwd(const void* gate, const void* up, void* out, long long rows, int I, int ld_gate, int ld_up,
                    int ld_out, int seg_rows, const int* seg_valid, hipStream_t stream);
int lmod_swiglu_bwd(const void* dact, const void* gate, const void* up, void* dgate, void* dup, long long rows, int I,
                    int ld_dact, int ld_gate, int ld_up, int ld_dgate, int ld_dup, int seg_rows, const int* seg_valid,
                    hipStream_t stream);
/* nn.GELU() of the mlp2x_gelu projector (multimodal_projector/builder.py:57-61). */
int lmod_gelu_fwd(const void* x, void* y, long long n, hipStream_t stream);
int lmod_gelu_bwd(const void* dy, const void* x, void* dx, long long n, hipStream_t stream);
int lmod_add_bf16(const void* a, const void* b, void* out, long long n, hipStream_t stream);
/* out[r] = idx[r] >= 0 ? srcA[idx[r]] : idx[r] <= -2 ? srcB[-(idx[r]+2)] : 0.  Embedding lookup +
 * image-feature splice of prepare_inputs_labels_for_multimodal (llava_arch.py:236-318), the MoE
 * dispatch (einsum 'sec,sm->ecm' of deepspeed MOELayer.forward), and their backward gathers. */
int lmod_gather_rows(const void* srcA, const void* srcB, const int* idx, void* out, long long rows, int H,
                     hipStream_t stream);
/* CLIP patch embedding front end (HF CLIPVisionEmbeddings; call site clip_encoder.py:54). */
int lmod_im2col_patch(const void* pixels, void* out, int B, int image_size, int patch, int Kpad, hipStream_t stream);
int lmod_vit_embed(const void* patch_emb, const void* cls, const void* pos, void* out, int B, int n_patches, int D,
                   hipStream_t stream);
/* torch.optim.AdamW step (HF `adamw_torch`, config/args.py:78) on fp32 master weights with a bf16
 * working copy; grad is fp32, scaled by grad_scale first.  zero_grad != 0: the gradient is cleared in the same pass
 * (the optimizer's zero_grad(), without a separate memset of the gradient buffer).  dev_scale (nullable): one fp32 on
 * the device that multiplies grad_scale — the gradient-clipping coefficient of lmod_clip_coef, read without a host sync. */
int lmod_adamw_step(float* master, void* param_bf16, float* grad, float* m, float* v, long long n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad,
                    const float* dev_scale, hipStream_t stream);
/* Global gradient-norm clipping (HF Trainer max_grad_norm = 1.0 under DeepSpeed `"gradient_clipping": "auto"`,
 * config/dpconfig/zero2_offload.json; torch.nn.utils.clip_grad_norm_ arithmetic).
 * lmod_sumsq_f32: out[0] (+)= sum(x[i]^2), deterministic (fixed reduction order); partials: >= 1024 floats of scratch.
 * lmod_clip_coef: coef[0] = min(1, max_norm / (sqrt(sumsq[0]) * norm_scale + 1e-6)); norm_out (nullable) gets the norm. */
int lmod_sumsq_f32(const float* x, long long n, float* partials, float* out, int accumulate, hipStream_t stream);
int lmod_clip_coef(const float* sumsq, float norm_scale, float max_norm, float* coef, float* norm_out, hipStream_t stream);
/* fp32 <-> bf16 (round to nearest even) over n elements: gradients are exchanged in bf16 like the reference's bf16
 * engine (DeepSpeed bf16 + ZeRO-2 reduce-scatters bf16 gradients). */
int lmod_cast_f32_bf16(const void* src, void* dst, long long n, int to_bf16, hipStream_t stream);

/* ---- attention -------------------------------------------------------------------------------
 * softmax(Q K^T * scale + mask) V with causal and right-padding masks, GQA (nh % nkv == 0).
 * Replaces F.scaled_dot_product_attention / flash_attn (qwen2/modeling_qwen2.py:700-708,535-581)
 * and CLIP encoder attention.  hd in {64, 128}.  lse: [B, nh, S] fp32 (may be NULL in fwd).
 * cu_seqlens (nullable, hd 128 only): UNPADDED layout — the samples' tokens are packed back to back, sample b occupies rows
 * cu_seqlens[b] .. cu_seqlens[b+1]-1 and S is the longest sample (lse / delta keep the [B, nh, S] layout). */
int lmod_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, const int* seqlens,
                  const int* cu_seqlens, int B, int S, int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
                  int causal, hipStream_t stream);
int lmod_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                  float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S, int nh, int nkv,
                  int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale,
                  int causal, hipStream_t stream);
/* lmod_attn_bwd with the gradient map of the rotary embedding (apply_rotary_pos_emb, qwen2/modeling_qwen2.py:146-171) applied to dQ and
 * dK in the backward kernels' epilogues, head dim 128 only: bit-identical to lmod_attn_bwd followed by lmod_rope(backward = 1) on the
 * q and k column blocks.  cos_t / sin_t: [max_pos x 128] bf16, pos: int32 per token row. */
int lmod_attn_bwd_rope(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                       float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                       int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                       float scale, int causal, const void* cos_t, const void* sin_t, const int* pos, hipStream_t stream);
/* The same backward (cos_t / sin_t / pos all NULL: lmod_attn_bwd; all set: lmod_attn_bwd_rope) with an optional workspace for the
 * HEAD-SPLIT form of the dK/dV kernel.  That kernel's grid is KV heads x key blocks x batch; grouped-query models with few KV heads
 * (the reference's d2s student Qwen2-0.5B, scripts/.../dense2sparse_distillation.sh:21: 14 heads over 2 KV heads) leave half of the
 * CUs without a workgroup.  lmod_attn_bwd_nsplit (host-side, no stream, no device work) returns into how many parts n every group
 * of query heads is cut for these shapes (1: no split); with split_ws_bytes >= n * 2 * B * S * nkv * hd * 4 (16-byte aligned) each
 * part runs as its own workgroup, stores fp32 partial sums into split_ws and a reduction kernel adds them in part order
 * (deterministic).  Without the workspace (NULL / too small), with fused RoPE or with cu_seqlens this is the unsplit launch.
 * The same workspace serves the dS-SPILL form (round 6) when the launch is not head-split: hd 128, no cu_seqlens, S % 256 == 0 and
 * split_ws_bytes >= B * nh * S * S * 2 (16-byte aligned) — the dK/dV kernel then also stores dS^T (bf16, [B * nh][S keys][S queries],
 * exactly the values its dK products consume) and dQ = scale * dS K (+ the fused RoPE gradient map) is ONE batched TN GEMM on the
 * weight-gradient kernel's K loop; the dQ kernel, which recomputes S, dP and the exponentials, is not launched (5 matmuls
 * instead of 7; the autograd of qwen2/modeling_qwen2.py:700-708).  Deterministic; dK / dV are bit-identical to the two-kernel form,
 * dQ differs by fp32 summation order only.  LMOD_ATTN_DS=0 (environment, read once) keeps the two-kernel form. */
int lmod_attn_bwd_nsplit(int B, int S, int nh, int nkv, int hd, int causal);
int lmod_attn_bwd_split(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                        float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                        int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                        float scale, int causal, const void* cos_t, const void* sin_t, const int* pos, void* split_ws,
                        long long split_ws_bytes, hipStream_t stream);

/* Single-query attention against a KV cache (generation: llava_qwen2_moe.py:453-473 prepare_inputs_for_generation,
 * qwen2/modeling_qwen2.py:290-309 with q_len 1).  q [B, ldq] (head h at column h*hd), caches [B, smax, ld_cache] (kv head
 * at column hk*hd, keys 0 .. lens[b]-1 valid, the new token's key/value already appended), out [B, ldo]. */
int lmod_attn_decode(const void* q, const void* kcache, const void* vcache, const int* lens, void* out, int B, int nh,
                     int nkv, int hd, int smax, int ldq, int ld_cache, int ldo, float scale, hipStream_t stream);

/* ---- multimodal splice / loss-row plans built on the device ----------------------------------------
 * prepare_inputs_labels_for_multimodal (llava_arch.py:155-334) as index maps, from int tensors already on the device.
 * lmod_splice_count: per sample the spliced length (pads stripped by attention_mask, every -200 token replaced by
 *   n_patches rows, cut at max_length if > 0) and the number of image slots it consumes (max(#-200, 1), :238-246).
 * lmod_splice_fill (S = max spliced length, read back by the host): idx[B*S] (>= 0: embedding row; <= -2: projector
 *   row -(idx+2); -1: right padding), new_labels[B*S] (-100 over image rows and padding), new_mask[B*S] (nullable),
 *   inv_idx[n_images_total * n_patches] (position of every projector row, -1 if cut; caller pre-fills with -1).
 * lmod_lossplan_*: rows of the [B*S] hidden states that carry loss (KD row t: labels[t] != -100, or every row when
 *   all_tokens — align_trainer.py:516-522; CE row t: labels[t+1] != -100 — llava_qwen2_moe.py:413-421), compacted in
 *   sample-major order: row_idx[R], inv_row_idx[B*S], kd_w/ce_w[R], ce_label[R] (-1 where no CE), seg_off[B+1], seg_id[R];
 *   R = sum(counts) is read back by the host between the two calls.  attention_mask: bytes (bool). */
int lmod_splice_count(const long long* input_ids, const unsigned char* attention_mask, int B, int T, int n_patches,
                      int max_length, int* lens, int* n_images, hipStream_t stream);
int lmod_splice_fill(const long long* input_ids, const unsigned char* attention_mask, const long long* labels, int B, int T,
                     int n_patches, int S, const int* lens, const int* n_images, int* idx, long long* new_labels,
                     unsigned char* new_mask, int* inv_idx, hipStream_t stream);
int lmod_lossplan_count(const long long* labels, int B, int S, int kd_rows, int ce_rows, int all_tokens, int* counts,
                        hipStream_t stream);
int lmod_lossplan_fill(const long long* labels, int B, int S, int kd_rows, int ce_rows, int all_tokens, const int* counts,
                       int* row_idx, int* inv_row_idx, float* kd_w, float* ce_w, int* ce_label, int* seg_off, int* seg_id,
                       hipStream_t stream);

/* Greedy decoding: out[r] = argmax_v logits[r, v] over bf16 rows (first maximum wins). */
int lmod_row_argmax_bf16(const void* logits, long long ld, int V, int* out, int R, hipStream_t stream);

/* ---- sparse MoE (DeepSpeed 0.9.5 TopKGate/top1gating/top2gating/MOELayer semantics) ------------
 * logits[T,E] = x.float() @ wg^T  (TopKGate.forward; wg kept fp32).
 * Experts per layer: 1 <= E <= 32 (`--num_experts`, config/args.py:46); the per-token kernels are instantiated for
 * ME = 8 / 16 / 32 expert slots and picked by E. */
int lmod_moe_router_fwd(const void* x, const float* wg, float* logits, int T, int H, int E, hipStream_t stream);
/* top-k (k in {1,2}) gating with capacity C: token-order slot assignment, drops, renormalised
 * combine weights, l_aux, exp_counts, slots_used[E] (live rows per capacity slab).
 * k = 2 (top2gating): `noise` [T,E] is added to the logits for the 2nd choice (NULL: none); noise_mode 1 draws
 *   Gumbel(0,1) noise in the kernel (Philox4x32-10 keyed by seed, counter = offset + token) — gumbel_rsample.
 * k = 1 (top1gating): `noise` [T,E] holds the random-token-selection priorities (use_rts=True: per expert the C tokens
 *   with the largest priority keep their slot, survivors numbered in token order; NULL: token order, use_rts=False);
 *   noise_mode 2 draws U(0,1) priorities in the kernel.
 * noise_out (nullable, [T,E]) receives the drawn noise.  scratch: 4*T + 3*ME*ceil(T/512) int32, ME = 8 / 16 / 32 for
 *   E <= 8 / 16 / 32. */
int lmod_moe_gate(const float* logits, const float* noise, int T, int E, int k, int C, float* gates, int* idx1,
                  int* idx2, int* slot1, int* slot2, float* w1, float* w2, int* slot_token, float* slot_w,
                  int* exp_counts, float* gate_sum, float* l_aux, int* slots_used, int* scratch, int noise_mode,
                  unsigned long long seed, unsigned long long offset, float* noise_out, hipStream_t stream);
/* out[t] = bf16(w1)*y[slot1[t]] + bf16(w2)*y[slot2[t]]   (einsum 'sec,ecm->sm'). */
int lmod_moe_combine_fwd(const void* y, const int* slot1, const int* slot2, const float* w1, const float* w2,
                         void* out, int T, int H, hipStream_t stream);
int lmod_moe_combine_bwd(const void* dout, const void* y, const int* slot1, const int* slot2, const int* slot_token,
                         const float* slot_w, void* dy, float* dw1, float* dw2, int T, int S, int H,
                         hipStream_t stream);
int lmod_moe_gate_bwd(const float* gates, const int* idx1, const int* idx2, const int* slot1, const int* slot2,
                      const float* dw1, const float* dw2, const int* exp_counts, const float* d_laux, float* dlogits,
                      int T, int E, int k, hipStream_t stream);
int lmod_moe_dispatch_bwd(const void* d_in, const int* slot1, const int* slot2, const float* dlogits, const float* wg,
                          void* dx, int T, int H, int E, hipStream_t stream);
/* dwg[E,H] (+)= dlogits^T x (fp32 router weight gradient).  workspace: ceil(T/64) * E * H floats.  H % 8 == 0. */
int lmod_moe_router_wgrad(const void* x, const float* dlogits, float* dwg, float* workspace, int T, int H, int E,
                          int accumulate, hipStream_t stream);

/* Residual-MoE (deepspeed.moe.layer.MoE(use_residual=True); flag at config/args.py:54-56, passed through at
 * llava_qwen2_moe.py:536-546): out = moe_out * c0 + mlp_out * c1 with (c0, c1) = softmax(coef_logits + coef_bias), the
 * 2-way `coefficient` Linear(hidden, 2); roundings follow the bf16 module.  p[T,2] (fp32) is saved for the backward,
 * which returns d_moe, d_mlp (bf16) and d_coef_logits[T,2] (fp32).  coef_logits come from lmod_moe_router_fwd (E = 2),
 * the coefficient weight gradient from lmod_moe_router_wgrad, its input gradient from lmod_small_linear_dgrad:
 * dx[T,H] (bf16) = dlogits[T,E] @ w[E,H] (fp32, E <= 32, H % 8 == 0). */
int lmod_moe_residual_mix_fwd(const void* moe_out, const void* mlp_out, const float* coef_logits, const float* coef_bias,
                              void* out, float* p, int T, int H, hipStream_t stream);
int lmod_moe_residual_mix_bwd(const void* dout, const void* moe_out, const void* mlp_out, const float* p, void* d_moe,
                              void* d_mlp, float* d_coef_logits, int T, int H, hipStream_t stream);
int lmod_small_linear_dgrad(const float* dlogits, const float* w, void* dx, int T, int H, int E, hipStream_t stream);

/* ---- distillation losses -----------------------------------------------------------------------
 * One pass per logits row.  stats[R][8] = {lse_s_full, lse_s_align, lse_t_align, x_kd, ce, s[label],
 * finite_mass, 0}:  x_kd = sum_v softmax(t[:Va])_v * log_softmax(s[:Va])_v with the isinf mask
 * (AlignTrainer.get_p/get_logp/compute_align_loss, align_trainer.py:455-528);  ce = lse_full - s[label]
 * (CrossEntropyLoss of llava_qwen2_moe.py:407-421; DPOTrainer.get_logp per-token logp = -ce,
 * dpo_trainer.py:483-495).  t may be NULL (no teacher row). */
int lmod_rowloss_fwd(const void* s, long long ld_s, int Vs, const void* t, long long ld_t, int Va, const int* label,
                     float* stats, int R, hipStream_t stream);
/* ds = kd_w[r]*kd_scale[seg]*(softmax_a(s)-softmax_a(t))[v<Va] + ce_w[r]*ce_scale[seg]*(softmax(s)-onehot). */
int lmod_rowloss_bwd(const void* s, long long ld_s, int Vs, const void* t, long long ld_t, int Va, const int* label,
                     const float* stats, const float* kd_w, const float* ce_w, const int* seg_id,
                     const float* kd_scale, const float* ce_scale, void* ds, long long ld_ds, int R,
                     hipStream_t stream);
/* out_sum[b] = sum_{r in [seg_off[b], seg_off[b+1])} w[r]*val[r*stride+col]; out_w[b] = sum w[r]. */
int lmod_segment_wsum(const float* val, int stride, int col, const float* w, const int* seg_off, int nseg,
                      float* out_sum, float* out_w, hipStream_t stream);
/* Materialising forms of AlignTrainer.get_p / get_logp (align_trainer.py:473-475,497-499): fp32 logits ->
 * fp32 softmax / log_softmax of the first Va columns; and the masked product-sum of compute_align_loss
 * (:509-514).  Slow-path API parity only; training uses lmod_rowloss_*. */
int lmod_row_softmax_f32(const float* logits, long long ld, int Va, int log_flag, float* out, int R,
                         hipStream_t stream);
int lmod_rowdot_masked(const float* p, const float* logp, int V, int R, float* x, hipStream_t stream);
/* DPOTrainer.dpo_loss (dpo_trainer.py:497-562), forward + d(mean loss)/d(policy logps).
 * loss_type: 0 sigmoid, 1 hinge, 2 ipo, 3 kto_pair (losses has 2B entries). */
int lmod_dpo_loss(const float* policy_chosen, const float* policy_rejected, const float* ref_chosen,
                  const float* ref_rejected, int B, float beta, float label_smoothing, int loss_type, float* losses,
                  float* chosen_rewards, float* rejected_rewards, float* d_policy_chosen, float* d_policy_rejected,
                  hipStream_t stream);

/* ---- collectives of the data-parallel step (RCCL over xGMI; bound at run time, no link dependency) -------------------
 * Replace what DeepSpeed's engine does for the reference around each optimizer step (train/align_trainer.py:326-434 builds it
 * from config/dpconfig/zero2*.json: gradient reduce-scatter / all-reduce, parameter all-gather) and the two all-to-alls of
 * deepspeed.moe.sharded_moe.MOELayer.forward (call site llava_qwen2_moe.py:536-546).  One communicator per process / GPU.
 * The Python package drives the same exchanges through torch.distributed; these are the boundary for other hosts. */
/* rank 0: fill the 128-byte communicator id; the host distributes it to the other ranks. */
int lmod_comm_unique_id(void* id128);
/* collective over all `world` ranks (current HIP device = this rank's GPU); *comm receives the handle. */
int lmod_comm_init(void** comm, const void* id128, int rank, int world);
int lmod_comm_destroy(void* comm);
/* buf[0..n) <- SUM over ranks, in place.  dtype: 0 fp32, 1 bf16.  (The mean's 1/world goes into lmod_adamw_step's grad_scale.) */
int lmod_allreduce_grads(void* comm, void* buf, long long n, int dtype, hipStream_t stream);
/* ZeRO-2 gradient phase: span[0 .. world*n_per_rank) -> chunk `rank` of the SUM, in place at span + rank*n_per_rank. */
int lmod_reduce_scatter_grads(void* comm, void* span, long long n_per_rank, int dtype, hipStream_t stream);
/* ZeRO-2 parameter phase: every rank publishes its updated chunk (span + rank*n_per_rank) to all, in place. */
int lmod_allgather_params(void* comm, void* span, long long n_per_rank, int dtype, hipStream_t stream);
/* Expert-parallel exchange of PACKED live rows (bf16, H wide): send_rows[d] consecutive rows of `send` go to peer d, recv_rows[s]
 * rows from peer s land consecutively in `recv` (HOST arrays of `world` counts).  One ncclSend / ncclRecv per peer in one group. */
int lmod_moe_all_to_all(void* comm, const void* send, void* recv, const long long* send_rows, const long long* recv_rows,
                        int H, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LMOD_HIP_H */
