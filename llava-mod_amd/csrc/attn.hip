// Attention entry points of the C-ABI (lmod_attn_fwd / _bwd / _bwd_rope / _bwd_split / _bwd_nsplit / _decode) on gfx950, bf16 in /
// fp32 accumulate.  Replaces F.scaled_dot_product_attention / flash_attn_func / the eager path of the reference decoder
// (qwen2/modeling_qwen2.py:700-708, :535-581, :290-309) and HF CLIP's encoder attention (call site
// multimodal_encoder/clip_encoder.py:54).  Causal + right-padding key mask (keys >= seqlens[b] are masked, like the 4-D mask of
// :1019-1027); GQA via `group`.
//
// The kernels live in attn_fwd2.hip (forward, hd 128 / 64) and attn_bwd2.hip (backward, hd 128 / 64); this file holds argument
// checks, dispatch, the small passes around them (delta = rowsum(dO * O), the head-split reduction) and the decode kernel.
// The round-1 generic kernels and the round-5 one-wave-per-SIMD forward are LAB arms (attn_lab.hip, attn_fwd3.hip; `make LAB=1`):
// in the product build a request for them (LMOD_ATTN_FWD=1|3, LMOD_ATTN_BWD=1, LMOD_ATTN_BWD64=1) returns LMOD_EUNSUPPORTED.
#include "attn_common.h"
#include <stdlib.h>
#ifndef LMOD_LAB
#define LMOD_LAB 0
#endif


// ============================================================================ backward: delta = rowsum(dO * O)
// 16 lanes per (token, head): 4 pairs per wave (one pair per wave left 48 of 64 lanes idle at hd 128)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                        float* __restrict__ delta, int B, int S, int nh, int HD,
                                                        int lddo, int ldo, const int* __restrict__ cu) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, c = lane & 15;
  const long long id = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;   // (b, s, h)
  const long long total = (long long)B * S * nh;
  float a = 0.f;
  long long tok = 0; int h = 0;
  const bool live = id < total;
  bool valid = live;
  if (live) {
    h = (int)(id % nh);
    tok = id / nh;                           // (b, s) with s < S; varlen: packed row cu[b] + s, skipped past the sample's end
  }
  long long row = tok;
  if (live && cu) {
    const int bb = (int)(tok / S), ss = (int)(tok % S);
    valid = ss < cu[bb + 1] - cu[bb];
    row = (long long)cu[bb] + ss;
  }
  if (valid) {
    for (int cidx = c; cidx < (HD >> 3); cidx += 16) {
      const u32x4 x = *(const u32x4*)(dO + row * lddo + h * HD + cidx * 8);
      const u32x4 y = *(const u32x4*)(O + row * ldo + h * HD + cidx * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) a += bflo(x[k]) * bflo(y[k]) + bfhi(x[k]) * bfhi(y[k]);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (live && c == 0) {
    const int b = (int)(tok / S), sidx = (int)(tok % S);
    delta[((long long)b * nh + h) * S + sidx] = a;
  }
}

template <typename KT>
static void set_lds(KT kern, int bytes) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static int check_common(int B, int S, int nh, int nkv, int hd, int ldq, int ldk, int ldv) {
  if (B <= 0 || S <= 0 || nh <= 0 || nkv <= 0 || nh % nkv) return LMOD_EINVAL;
  if (hd != 64 && hd != 128) return LMOD_EUNSUPPORTED;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || ldq < nh * hd || ldk < nkv * hd || ldv < nkv * hd) return LMOD_EINVAL;
  if ((long long)S * ldk * 2 >= 0x7fffffffLL || (long long)S * ldq * 2 >= 0x7fffffffLL || (long long)S * ldv * 2 >= 0x7fffffffLL)
    return LMOD_EUNSUPPORTED;
  return LMOD_OK;
}

extern "C" {

// Q [B*S, ldq] (head h at column h*hd), K/V [B*S, ldk/ldv] (kv head at column hk*hd); O [B*S, ldo];
// lse [B, nh, S] fp32 (natural log of the scaled-score partition function; may be NULL).
// seqlens [B] i32 or NULL: keys >= seqlens[b] are masked.
#ifndef LMOD_ATTN_FWD_DEFAULT
#define LMOD_ATTN_FWD_DEFAULT 2
#endif
static int xcd_remap_on() {    // LMOD_ATTN_XCD=0: hardware workgroup ids as they come, also for head counts that are no multiple of 8 (A/B timing)
  static const int v = [] { const char* e = getenv("LMOD_ATTN_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}
int lmod_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, const int* seqlens,
                  const int* cu_seqlens, int B, int S, int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
                  int causal, hipStream_t stream) {
  if (!Q || !K || !V || !O) return LMOD_EINVAL;
  int rc = check_common(B, S, nh, nkv, hd, ldq, ldk, ldv);
  if (rc) return rc;
  if ((ldo & 7) || ldo < nh * hd) return LMOD_EINVAL;
  AttnP p = {};
  p.xcd_remap = xcd_remap_on();
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O; p.LSE = lse;
  p.seqlens = seqlens; p.cu = cu_seqlens; p.B = B; p.S = S; p.nh = nh; p.group = nh / nkv;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale = scale;
  if (cu_seqlens && (hd != 128 || seqlens)) return LMOD_EUNSUPPORTED;      // packed (varlen) layout: the hd-128 kernels only
  // LMOD_ATTN_FWD (read once): 2 = the 8-wave 32x32x16 kernel (attn_fwd2.hip; the only forward of the product build).  LAB builds
  // also carry 3 = the one-wave-per-SIMD kernel of round 5 (attn_fwd3.hip: 4 waves x 64 queries, hd 128) and 1 = the round-1
  // 16x16x32 kernel (attn_lab.hip); asking the product build for one of them fails loudly instead of silently timing the default.
  static const int fwd_ver = [] { const char* e = getenv("LMOD_ATTN_FWD"); return e ? atoi(e) : LMOD_ATTN_FWD_DEFAULT; }();
#if LMOD_LAB
  if (hd == 128 && fwd_ver == 3) { lmod_launch_attn_fwd3(p, causal, stream); return lmod_launch_status(); }
  if (fwd_ver == 1 && !cu_seqlens) { lmod_launch_attn_fwd_generic(p, causal, stream, hd); return lmod_launch_status(); }
#else
  if (fwd_ver == 1 || fwd_ver == 3) return LMOD_EUNSUPPORTED;
#endif
  lmod_launch_attn_fwd2(p, causal, stream, hd);
  return lmod_launch_status();
}

// delta_ws: [B, nh, S] fp32 workspace.  dQ/dK/dV use the same layouts as Q/K/V (own leading dims).
static bool bwd_generic() {      // LMOD_ATTN_BWD=1: hd-128 backward through the generic 16x16x32 kernels (LAB builds: A/B timing)
  static const bool v = [] { const char* e = getenv("LMOD_ATTN_BWD"); return e && e[0] == '1'; }();
  return v;
}

#ifndef LMOD_ATTN_BWD64_DEFAULT
#define LMOD_ATTN_BWD64_DEFAULT 2
#endif
static bool bwd_nosplit() {      // LMOD_ATTN_BWD_SPLIT=0: never head-split the dK/dV launch (A/B timing)
  static const bool v = [] { const char* e = getenv("LMOD_ATTN_BWD_SPLIT"); return e && e[0] == '0'; }();
  return v;
}
static int attn_cus() {
  static const int v = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    return n;
  }();
  return v;
}
static bool bwd64_fast() {       // LMOD_ATTN_BWD64=1: hd-64 backward through the generic kernels; 2: the one-wave-per-SIMD kernels
  static const bool v = [] { const char* e = getenv("LMOD_ATTN_BWD64"); return (e && e[0] ? atoi(e) : LMOD_ATTN_BWD64_DEFAULT) != 1; }();
  return v;
}

// Head-split dK/dV launches: dK = bf16(scale * sum over parts), dV = bf16(sum over parts), parts added in index order (deterministic).
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const float* __restrict__ ws, int nsplit, long long rows, int width,
                                                               bf16_t* __restrict__ dK, int lddk, bf16_t* __restrict__ dV, int lddv,
                                                               float scale) {
  const int per_row = width >> 3;                                  // 8 features per thread
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x, n = rows * per_row;
  if (id >= 2 * n) return;
  const int which = id >= n;
  const long long e = which ? id - n : id, row = e / per_row;
  const int col = (int)(e - row * per_row) * 8;
  const long long plane = rows * width;
  const float* src = ws + which * plane + row * width + col;
  f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
  for (int s = 1; s < nsplit; ++s) {
    a += *(const f32x4*)(src + 2 * s * plane);
    b += *(const f32x4*)(src + 2 * s * plane + 4);
  }
  const float m = which ? 1.f : scale;
  const u32x4 o = {pack2bf(a[0] * m, a[1] * m), pack2bf(a[2] * m, a[3] * m), pack2bf(b[0] * m, b[1] * m), pack2bf(b[2] * m, b[3] * m)};
  *(u32x4*)((which ? dV + row * lddv : dK + row * lddk) + col) = o;
}

// How many parts lmod_attn_bwd_split cuts every KV head's query-head group into for these shapes (1: no split).  The dK/dV kernel's
// grid is KV heads x owner blocks (causal: pairs) x batch; with few KV heads (Qwen2-0.5B: 2) that is fewer workgroups than CUs.
int lmod_attn_bwd_nsplit(int B, int S, int nh, int nkv, int hd, int causal) {
  if (B < 1 || S < 1 || nkv < 1 || nh < nkv || nh % nkv || (hd != 64 && hd != 128)) return 1;
  if (bwd_generic() || (hd == 64 && !bwd64_fast()) || bwd_nosplit()) return 1;
  const int nb = (S + 255) / 256, gx = causal ? (nb + 1) / 2 : nb, group = nh / nkv;
  const long long wgs = (long long)nkv * gx * B;
  int n = 1;
  while (n < group && n < 8 && wgs * n * 10 < 9LL * attn_cus()) ++n;
  // every part must own a query head: with per = ceil(group / n) heads per part, ceil(group / per) parts are non-empty
  // (group 7, n 5 -> per 2 -> 4 parts; a fifth would run its prologue, store zeros and grow the workspace for nothing)
  const int per = (group + n - 1) / n;
  return (group + per - 1) / per;
}

static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                         float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                         int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                         float scale, int causal, const void* rope_cos, const void* rope_sin, const int* rope_pos,
                         void* split_ws, long long split_ws_bytes, hipStream_t stream) {
  if (!Q || !K || !V || !O || !dO || !lse || !delta_ws || !dQ || !dK || !dV) return LMOD_EINVAL;
  if (rope_pos && (!rope_cos || !rope_sin)) return LMOD_EINVAL;
  if (rope_pos && (hd != 128 || bwd_generic())) return LMOD_EUNSUPPORTED;     // fused only in the hd-128 kernels (attn_bwd2.hip)
  int rc = check_common(B, S, nh, nkv, hd, ldq, ldk, ldv);
  if (rc) return rc;
  if ((ldo & 7) || (lddo & 7) || (lddq & 7) || (lddk & 7) || (lddv & 7) || ldo < nh * hd || lddo < nh * hd ||
      lddq < nh * hd || lddk < nkv * hd || lddv < nkv * hd) return LMOD_EINVAL;
  if ((long long)S * lddo * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  AttnP p = {};
  p.xcd_remap = xcd_remap_on();
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O; p.LSE = (float*)lse;
  p.dO = (const bf16_t*)dO; p.Delta = delta_ws; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV;
  p.seqlens = seqlens; p.cu = cu_seqlens; p.B = B; p.S = S; p.nh = nh; p.group = nh / nkv;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.scale = scale;
  p.rope_cos = (const bf16_t*)rope_cos; p.rope_sin = (const bf16_t*)rope_sin; p.rope_pos = rope_pos;
  if (cu_seqlens && (hd != 128 || seqlens)) return LMOD_EUNSUPPORTED;
  const long long rows = (long long)B * S * nh;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, (const bf16_t*)dO,
                     (const bf16_t*)O, delta_ws, B, S, nh, hd, lddo, ldo, cu_seqlens);
  if ((hd == 128 || (hd == 64 && bwd64_fast())) && !bwd_generic()) {
    const int ns = (split_ws && !rope_pos && !cu_seqlens) ? lmod_attn_bwd_nsplit(B, S, nh, nkv, hd, causal) : 1;
    if (ns > 1 && split_ws_bytes >= (long long)ns * 2 * B * S * nkv * hd * 4) {
      if ((reinterpret_cast<uintptr_t>(split_ws) & 15) != 0) return LMOD_EINVAL;
      p.split_ws = (float*)split_ws; p.nsplit = ns; p.split_rows = (long long)B * S;
    }
    // dS spill (round 6): with a workspace of B * nh * S * S bf16 the dK/dV kernel also stores dS^T and dQ = scale * dS K becomes one batched
    // TN GEMM (lmod_launch_attn_dq_gemm); the dQ kernel — S, dP and the exponentials a second time, 3 of the two-kernel form's 7
    // matmuls — is not launched.  hd 128, dense layout (no cu_seqlens), whole 256-row blocks, unsplit launches; LMOD_ATTN_DS=0 keeps
    // the two-kernel form (A/B arm, read once).
    static const bool ds_on = [] { const char* e = getenv("LMOD_ATTN_DS"); return !(e && e[0] == '0'); }();
    if (ds_on && hd == 128 && !cu_seqlens && (S & 255) == 0 && p.nsplit <= 1 && split_ws &&
        (reinterpret_cast<uintptr_t>(split_ws) & 15) == 0 && split_ws_bytes >= (long long)B * nh * S * S * 2 &&
        (long long)S * S * 2 < 0x7fffffffLL) {
      p.ds_ws = (bf16_t*)split_ws;
      lmod_launch_attn_bwd2(p, causal, stream, hd);        // delta + the dK/dV kernel (spilling variant)
      lmod_launch_attn_dq_gemm(p, causal, stream);
      return lmod_launch_status();
    }
    lmod_launch_attn_bwd2(p, causal, stream, hd);
    if (p.nsplit > 1) {
      const int width = nkv * hd;
      const long long thr = 2 * p.split_rows * (width >> 3);
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, stream, p.split_ws, p.nsplit,
                         p.split_rows, width, p.dK, p.lddk, p.dV, p.lddv, p.scale);
    }
    return lmod_launch_status();
  }
#if LMOD_LAB
  lmod_launch_attn_bwd_generic(p, causal, stream, hd);
  return lmod_launch_status();
#else
  return LMOD_EUNSUPPORTED;        // LMOD_ATTN_BWD=1 / LMOD_ATTN_BWD64=1 name kernels this build does not contain (attn_lab.hip)
#endif
}

int lmod_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                  float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                  int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                  float scale, int causal, hipStream_t stream) {
  return attn_bwd_impl(Q, K, V, O, dO, lse, delta_ws, dQ, dK, dV, seqlens, cu_seqlens, B, S, nh, nkv, hd, ldq, ldk, ldv, ldo, lddo,
                       lddq, lddk, lddv, scale, causal, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// lmod_attn_bwd with the gradient map of the rotary embedding applied to dQ and dK before they are stored (head dim 128):
// what autograd does for apply_rotary_pos_emb (qwen2/modeling_qwen2.py:146-171) after the attention backward, without the extra
// pass over the d(QKV) buffer.  pos: int32 per token row of Q / K (the same indexing as dQ / dK), tables [max_pos, 128] bf16.
int lmod_attn_bwd_rope(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                       float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                       int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                       float scale, int causal, const void* cos_t, const void* sin_t, const int* pos, hipStream_t stream) {
  if (!cos_t || !sin_t || !pos) return LMOD_EINVAL;
  return attn_bwd_impl(Q, K, V, O, dO, lse, delta_ws, dQ, dK, dV, seqlens, cu_seqlens, B, S, nh, nkv, hd, ldq, ldk, ldv, ldo, lddo,
                       lddq, lddk, lddv, scale, causal, cos_t, sin_t, pos, nullptr, 0, stream);
}

// lmod_attn_bwd (cos_t / sin_t / pos all NULL) or lmod_attn_bwd_rope (all set) with an optional workspace for the head-split form of
// the dK/dV kernel: split_ws_bytes >= lmod_attn_bwd_nsplit(...) * 2 * B * S * nkv * hd * 4, 16-byte aligned.  Without it (NULL, too
// small, a split count of 1, fused RoPE or cu_seqlens) this is the unsplit launch.
int lmod_attn_bwd_split(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                        float* delta_ws, void* dQ, void* dK, void* dV, const int* seqlens, const int* cu_seqlens, int B, int S,
                        int nh, int nkv, int hd, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv,
                        float scale, int causal, const void* cos_t, const void* sin_t, const int* pos, void* split_ws,
                        long long split_ws_bytes, hipStream_t stream) {
  if ((cos_t || sin_t || pos) && (!cos_t || !sin_t || !pos)) return LMOD_EINVAL;
  return attn_bwd_impl(Q, K, V, O, dO, lse, delta_ws, dQ, dK, dV, seqlens, cu_seqlens, B, S, nh, nkv, hd, ldq, ldk, ldv, ldo, lddo,
                       lddq, lddk, lddv, scale, causal, cos_t, sin_t, pos, split_ws, split_ws_bytes, stream);
}

}  // extern "C"

// ============================================================================ single-query (decode) attention
// Generation with a KV cache (reference: llava_qwen2_moe.py:453-473 prepare_inputs_for_generation + HF generate;
// qwen2/modeling_qwen2.py:290-309 with q_len == 1).  One workgroup per (batch sample, query head): the new token's
// query against lens[b] cached keys.  HBM-bound: K and V of the head are read once.
//   phase 1: thread <-> key (256 keys per chunk): full dot product with the query (q broadcast from LDS), scores to LDS,
//            running max / sum over chunks (online softmax, fp32);
//   phase 2: thread <-> (feature pair, key slice): p-weighted sum of the chunk's V rows, coalesced 4-byte loads.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                         const bf16_t* __restrict__ vc, const int* __restrict__ lens,
                                                         bf16_t* __restrict__ out, int nh, int group, int smax, int ldq,
                                                         int ldc, int ldo, float scale) {
  __shared__ float sq[HD];
  __shared__ float sp[256];
  __shared__ float red[8];
  __shared__ float oacc[256 / (HD / 2)][HD];      // one partial per key slice
  const int h = blockIdx.x, b = blockIdx.y, hk = h / group, tid = threadIdx.x;
  const int len = min(lens[b], smax);
  if (tid < HD) sq[tid] = bf2f(q[(long long)b * ldq + h * HD + tid]) * scale;
  __syncthreads();
  const bf16_t* kb = kc + (long long)b * smax * ldc + hk * HD;
  const bf16_t* vb = vc + (long long)b * smax * ldc + hk * HD;
  constexpr int DP = HD / 2;                 // feature pairs; thread owns pair tid % DP in key slice tid / DP
  constexpr int NSL = 256 / DP;              // key slices per chunk (4 at hd 128, 8 at hd 64)
  const int dp = tid % DP, sl = tid / DP;
  float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int key = c0 + tid;
    float s = -INFINITY;
    if (key < len) {
      const bf16_t* kr = kb + (long long)key * ldc;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 8) {
        const u32x4 w = *(const u32x4*)(kr + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) a += bflo(w[j]) * sq[d + 2 * j] + bfhi(w[j]) * sq[d + 2 * j + 1];
      }
      s = a;
    }
    const float cm = block_max<4>(s, red);
    const float mnew = fmaxf(m, cm);
    const float alpha = (mnew == -INFINITY) ? 1.f : __expf(m - mnew);
    const float p = (key < len) ? __expf(s - mnew) : 0.f;
    const float cs = block_sum<4>(p, red);
    l = l * alpha + cs;
    m = mnew;
    o0 *= alpha; o1 *= alpha;
    __syncthreads();
    sp[tid] = p;
    __syncthreads();
    const int nk = min(256, len - c0);
    for (int kk = sl; kk < nk; kk += NSL) {
      const uint32_t w = *(const uint32_t*)(vb + (long long)(c0 + kk) * ldc + dp * 2);
      const float pv = sp[kk];
      o0 += pv * bflo(w); o1 += pv * bfhi(w);
    }
  }
  __syncthreads();
  float* oa = &oacc[0][0];                   // [NSL][HD] partial sums of the key slices (4 x 128 or 8 x 64 floats)
  oa[sl * HD + dp * 2] = o0; oa[sl * HD + dp * 2 + 1] = o1;
  __syncthreads();
  if (tid < HD) {
    float acc = 0.f;
#pragma unroll
    for (int x = 0; x < NSL; ++x) acc += oa[x * HD + tid];
    out[(long long)b * ldo + h * HD + tid] = f2bf(l > 0.f ? acc / l : 0.f);
  }
}

extern "C" int lmod_attn_decode(const void* q, const void* kcache, const void* vcache, const int* lens, void* out, int B,
                                int nh, int nkv, int hd, int smax, int ldq, int ld_cache, int ldo, float scale,
                                hipStream_t stream) {
  if (B < 0 || nh <= 0 || nkv <= 0 || nh % nkv || smax <= 0) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  if (!q || !kcache || !vcache || !lens || !out) return LMOD_EINVAL;
  if (hd != 64 && hd != 128) return LMOD_EUNSUPPORTED;
  if ((ld_cache & 7) || ld_cache < nkv * hd || ldq < nh * hd || ldo < nh * hd || ((uintptr_t)kcache & 15) || ((uintptr_t)vcache & 15))
    return LMOD_EINVAL;
  const dim3 grid(nh, B);
  if (hd == 128)
    hipLaunchKernelGGL(attn_decode_kernel<128>, grid, dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)kcache,
                       (const bf16_t*)vcache, lens, (bf16_t*)out, nh, nh / nkv, smax, ldq, ld_cache, ldo, scale);
  else
    hipLaunchKernelGGL(attn_decode_kernel<64>, grid, dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)kcache,
                       (const bf16_t*)vcache, lens, (bf16_t*)out, nh, nh / nkv, smax, ldq, ld_cache, ldo, scale);
  return lmod_launch_status();
}
