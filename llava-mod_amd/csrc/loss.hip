// Distillation-loss row kernels (gfx950).  One workgroup per logits row, ONE pass over the row
// with online (max, sum) pairs; replaces the reference's >= 6 full passes over fp32 [B,S,V]:
//   mimic  (align_trainer.py:473-475,497-499,503-528): softmax(t[:Va]) x log_softmax(s[:Va]),
//          isinf mask, vocab sum                                     -> x_kd[row]
//   LM CE  (llava_qwen2_moe.py:407-421, shifted by the caller's row selection)  -> ce[row]
//   DPO    (dpo_trainer.py:483-495): log_softmax over the FULL vocab gathered at the label
//          = -ce[row]; per-sample sums by lmod_segment_wsum.
// Logits arrive as the bf16 lm_head output (the reference's `.float()` of a bf16 tensor is exact),
// all arithmetic fp32.  Backward writes d(logits) in bf16 (what autograd hands back through
// `.float()`), optionally in place over the student logits.
#include "common.h"

#define NSTAT 8
// stats row: 0 lse_s_full, 1 lse_s_align, 2 lse_t_align, 3 x_kd, 4 ce, 5 s[label], 6 finite-mass, 7 unused

struct Online { float m, z; };
__device__ __forceinline__ void on_add(Online& o, float v) {
  if (v > o.m) { o.z = o.z * __expf(o.m - v) + 1.f; o.m = v; } else { o.z += __expf(v - o.m); }
}
__device__ __forceinline__ void on_merge(Online& a, const Online& b) {
  const float M = fmaxf(a.m, b.m);
  const float za = (a.m == -INFINITY) ? 0.f : a.z * __expf(a.m - M);
  const float zb = (b.m == -INFINITY) ? 0.f : b.z * __expf(b.m - M);
  a.m = M; a.z = za + zb;
}

template <int NW>
__device__ __forceinline__ Online block_online(Online o, float* red /* 2*NW floats */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Online b; b.m = __shfl_xor(o.m, off, 64); b.z = __shfl_xor(o.z, off, 64);
    on_merge(o, b);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[2 * w] = o.m; red[2 * w + 1] = o.z; }
  __syncthreads();
  Online r; r.m = red[0]; r.z = red[1];
#pragma unroll
  for (int i = 1; i < NW; ++i) { Online b; b.m = red[2 * i]; b.z = red[2 * i + 1]; on_merge(r, b); }
  return r;
}

__global__ __launch_bounds__(256) void rowloss_fwd_kernel(const bf16_t* __restrict__ s, long long ld_s, int Vs,
                                                         const bf16_t* __restrict__ t, long long ld_t, int Va,
                                                         const int* __restrict__ label, float* __restrict__ stats) {
  __shared__ float red[16];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* sr = s + row * ld_s;
  const bf16_t* tr = t ? t + row * ld_t : nullptr;
  Online os = {-INFINITY, 0.f};   // student, aligned slice [0, Va)
  Online ox = {-INFINITY, 0.f};   // student, extra slice [Va, Vs)
  Online ot = {-INFINITY, 0.f};   // teacher, aligned slice
  float dt = 0.f, ft = 0.f;       // sum exp(t - ot.m) * s  and  sum_{s finite} exp(t - ot.m)
  const int nca = Va >> 3, ncs = Vs >> 3;
  for (int c = tid; c < nca; c += 256) {
    const u32x4 sv = *(const u32x4*)(sr + c * 8);
    float sf[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sf[2 * k] = bflo(sv[k]); sf[2 * k + 1] = bfhi(sv[k]); }
    float cm = sf[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) cm = fmaxf(cm, sf[k]);
    if (cm > os.m) { os.z *= __expf(os.m - cm); os.m = cm; }
    if (os.m > -INFINITY) {
#pragma unroll
      for (int k = 0; k < 8; ++k) os.z += __expf(sf[k] - os.m);
    }
    if (tr) {
      const u32x4 tv = *(const u32x4*)(tr + c * 8);
      float tf[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { tf[2 * k] = bflo(tv[k]); tf[2 * k + 1] = bfhi(tv[k]); }
      float tm = tf[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) tm = fmaxf(tm, tf[k]);
      if (tm > ot.m) { const float f = __expf(ot.m - tm); ot.z *= f; dt *= f; ft *= f; ot.m = tm; }
      if (ot.m > -INFINITY) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float e = __expf(tf[k] - ot.m);
          ot.z += e;
          if (sf[k] > -INFINITY) { dt += e * sf[k]; ft += e; }   // isinf(logp) mask of the reference
        }
      }
    }
  }
  for (int c = nca + tid; c < ncs; c += 256) {
    const u32x4 sv = *(const u32x4*)(sr + c * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) { on_add(ox, bflo(sv[k])); on_add(ox, bfhi(sv[k])); }
  }
  // combine the teacher triple across the block at a common max
  float tM = block_max<4>(ot.m, red);
  const float tsc = (ot.m == -INFINITY) ? 0.f : __expf(ot.m - tM);
  const float tz = block_sum<4>(ot.z * tsc, red);
  const float td = block_sum<4>(dt * tsc, red);
  const float tf_ = block_sum<4>(ft * tsc, red);
  const Online bs = block_online<4>(os, red);
  const Online bx = block_online<4>(ox, red);
  if (tid == 0) {
    const float lse_a = bs.m + logf(bs.z);
    Online full = bs; on_merge(full, bx);
    const float lse_f = full.m + logf(full.z);
    float* st = stats + (long long)row * NSTAT;
    st[0] = lse_f; st[1] = lse_a;
    float xkd = 0.f, lse_t = 0.f, fm = 1.f;
    if (tr) {
      lse_t = tM + logf(tz);
      fm = tf_ / tz;
      xkd = td / tz - lse_a * fm;
    }
    st[2] = lse_t; st[3] = xkd; st[6] = fm; st[7] = 0.f;
    const int lb = label ? label[row] : -1;
    float sl = 0.f, ce = 0.f;
    if (lb >= 0 && lb < Vs) { sl = bf2f(sr[lb]); ce = lse_f - sl; }
    st[4] = ce; st[5] = sl;
  }
}

// ds[v] = ckd * (softmax_align(s)[v] - softmax_align(t)[v]) * [v < Va] + cce * (softmax_full(s)[v] - [v == label])
// ckd = kd_w[row] * kd_scale[seg], cce = ce_w[row] * ce_scale[seg]   (seg = seg_id ? seg_id[row] : 0)
__global__ __launch_bounds__(256) void rowloss_bwd_kernel(const bf16_t* s, long long ld_s, int Vs,
                                                         const bf16_t* __restrict__ t, long long ld_t, int Va,
                                                         const int* __restrict__ label, const float* __restrict__ stats,
                                                         const float* __restrict__ kd_w, const float* __restrict__ ce_w,
                                                         const int* __restrict__ seg_id, const float* __restrict__ kd_scale,
                                                         const float* __restrict__ ce_scale, bf16_t* ds, long long ld_ds) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* sr = s + row * ld_s;
  const bf16_t* tr = t ? t + row * ld_t : nullptr;
  bf16_t* dr = ds + row * ld_ds;
  const float* st = stats + (long long)row * NSTAT;
  const int sg = seg_id ? seg_id[row] : 0;
  const float ckd = (tr && kd_w && kd_scale) ? kd_w[row] * kd_scale[sg] : 0.f;
  const float cce = (ce_w && ce_scale) ? ce_w[row] * ce_scale[sg] : 0.f;
  const float lse_f = st[0], lse_a = st[1], lse_t = st[2];
  const int lb = label ? label[row] : -1;
  const int nca = Va >> 3, ncs = Vs >> 3;
  for (int c = tid; c < ncs; c += 256) {
    const u32x4 sv = *(const u32x4*)(sr + c * 8);
    float g[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { g[2 * k] = bflo(sv[k]); g[2 * k + 1] = bfhi(sv[k]); }
    float tf[8];
    const bool al = (c < nca) && (ckd != 0.f);
    if (al) {
      const u32x4 tv = *(const u32x4*)(tr + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) { tf[2 * k] = bflo(tv[k]); tf[2 * k + 1] = bfhi(tv[k]); }
    }
    u32x4 o;
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float acc = 0.f;
      if (al) acc += ckd * (__expf(g[k] - lse_a) - __expf(tf[k] - lse_t));
      if (cce != 0.f) acc += cce * (__expf(g[k] - lse_f) - ((c * 8 + k) == lb ? 1.f : 0.f));
      r[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(r[2 * k], r[2 * k + 1]);
    *(u32x4*)(dr + c * 8) = o;
  }
}

// out_sum[b] = sum_{r in [off[b], off[b+1])} w[r] * val[r*stride + col];  out_w[b] = sum w[r]
__global__ __launch_bounds__(256) void segment_wsum_kernel(const float* __restrict__ val, int stride, int col,
                                                          const float* __restrict__ w, const int* __restrict__ off,
                                                          float* __restrict__ out_sum, float* __restrict__ out_w) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const int lo = off[b], hi = off[b + 1];
  float a = 0.f, ws = 0.f;
  for (int r = lo + threadIdx.x; r < hi; r += 256) {
    const float wr = w ? w[r] : 1.f;
    a += wr * val[(long long)r * stride + col];
    ws += wr;
  }
  a = block_sum<4>(a, red);
  ws = block_sum<4>(ws, red);
  if (threadIdx.x == 0) { out_sum[b] = a; if (out_w) out_w[b] = ws; }
}

// DPO / KTO-pair loss on per-sample log-prob sums (dpo_trainer.py:497-562), forward + gradient.
// loss_type: 0 sigmoid, 1 hinge, 2 ipo, 3 kto_pair.  losses has B entries (2B for kto_pair).
// d_pc / d_pr = d(mean(losses)) / d(policy_{chosen,rejected}_logps).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float logsigmoidf_(float x) { return fminf(x, 0.f) - log1pf(__expf(-fabsf(x))); }

__global__ void dpo_loss_kernel(const float* __restrict__ pc, const float* __restrict__ pr, const float* __restrict__ rc,
                                const float* __restrict__ rr, int B, float beta, float ls, int loss_type,
                                float* __restrict__ losses, float* __restrict__ chosen_rewards,
                                float* __restrict__ rejected_rewards, float* __restrict__ d_pc, float* __restrict__ d_pr) {
  __shared__ float red[8];
  const int i = threadIdx.x;
  const bool on = i < B;
  const float a = on ? pc[i] : 0.f, b = on ? pr[i] : 0.f, c = on ? rc[i] : 0.f, d = on ? rr[i] : 0.f;
  if (on) { chosen_rewards[i] = beta * (a - c); rejected_rewards[i] = beta * (b - d); }
  if (loss_type != 3) {
    const float z = (a - b) - (c - d);
    float L = 0.f, dz = 0.f;
    if (loss_type == 0) {
      L = -logsigmoidf_(beta * z) * (1.f - ls) - logsigmoidf_(-beta * z) * ls;
      dz = -beta * sigmoidf_(-beta * z) * (1.f - ls) + beta * sigmoidf_(beta * z) * ls;
    } else if (loss_type == 1) {
      L = fmaxf(1.f - beta * z, 0.f);
      dz = (1.f - beta * z > 0.f) ? -beta : 0.f;
    } else {
      const float u = z - 1.f / (2.f * beta);
      L = u * u; dz = 2.f * u;
    }
    if (on) { losses[i] = L; d_pc[i] = dz / (float)B; d_pr[i] = -dz / (float)B; }
  } else {
    // chosen_KL = clamp(mean(pc - rc), 0); rejected_KL = clamp(mean(pr - rr), 0)
    const float mc = block_sum<4>(on ? (a - c) : 0.f, red) / (float)B;
    const float mr = block_sum<4>(on ? (b - d) : 0.f, red) / (float)B;
    const float ckl = fmaxf(mc, 0.f), rkl = fmaxf(mr, 0.f);
    const float s1 = sigmoidf_(beta * ((a - c) - rkl));    // losses[i]     = 1 - s1
    const float s2 = sigmoidf_(beta * (ckl - (b - d)));    // losses[B + i] = 1 - s2
    const float g1 = on ? -beta * s1 * (1.f - s1) : 0.f;   // dL1/d(arg1)
    const float g2 = on ? -beta * s2 * (1.f - s2) : 0.f;   // dL2/d(arg2)
    const float G1 = block_sum<4>(g1, red), G2 = block_sum<4>(g2, red);
    if (on) {
      losses[i] = 1.f - s1; losses[B + i] = 1.f - s2;
      const float n = 2.f * (float)B;                      // mean over the concatenated 2B vector
      // arg1_i = (pc_i - rc_i) - rkl(pr);  arg2_i = ckl(pc) - (pr_i - rr_i)
      float dpc = g1 + ((mc >= 0.f) ? G2 / (float)B : 0.f);   // torch clamp(min=0) passes grad at equality
      float dpr = -g2 - ((mr >= 0.f) ? G1 / (float)B : 0.f);
      d_pc[i] = dpc / n; d_pr[i] = dpr / n;
    }
  }
}


// ---- materialising helpers for the reference's get_p / get_logp / compute_align_loss API (slow path) ----
// out[r, :Va] = softmax(logits[r, :Va]) (log == 0) or log_softmax (log != 0); fp32 in, fp32 out.
__global__ __launch_bounds__(256) void row_softmax_f32_kernel(const float* __restrict__ x, long long ld, int Va, int lg,
                                                             float* __restrict__ out) {
  __shared__ float red[16];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + row * ld;
  float* o = out + (long long)row * Va;
  Online on = {-INFINITY, 0.f};
  for (int i = tid; i < Va; i += 256) on_add(on, xr[i]);
  const Online b = block_online<4>(on, red);
  const float lse = b.m + logf(b.z);
  for (int i = tid; i < Va; i += 256) o[i] = lg ? (xr[i] - lse) : __expf(xr[i] - lse);
}
// x[r] = sum_v (isinf(logp[r,v]) ? 0 : p[r,v] * logp[r,v])      (align_trainer.py:509-514)
__global__ __launch_bounds__(256) void rowdot_masked_kernel(const float* __restrict__ p, const float* __restrict__ lp,
                                                           int V, float* __restrict__ x) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  float a = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) {
    const float l = lp[(long long)row * V + i];
    if (!isinf(l)) a += p[(long long)row * V + i] * l;
  }
  a = block_sum<4>(a, red);
  if (threadIdx.x == 0) x[row] = a;
}

// ---------------------------------------------------------------- greedy decoding: argmax over a logits row
// (first maximum wins, like torch.argmax on ties in practice; NaN never wins)
__global__ __launch_bounds__(256) void row_argmax_kernel(const bf16_t* __restrict__ x, long long ld, int V, int* __restrict__ out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const bf16_t* r = x + (long long)blockIdx.x * ld;
  float best = -INFINITY; int idx = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float v = bf2f(r[c]);
    if (v > best) { best = v; idx = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[blockIdx.x] = (idx == 0x7fffffff) ? 0 : idx;
  }
}

extern "C" {

int lmod_rowloss_fwd(const void* s, long long ld_s, int Vs, const void* t, long long ld_t, int Va,
                     const int* label, float* stats, int R, hipStream_t stream) {
  if (!s || !stats || R < 0 || Vs <= 0 || (Vs & 7) || Va < 0 || (Va & 7) || Va > Vs || (ld_s & 7) ||
      (t && (ld_t & 7))) return LMOD_EINVAL;
  if (R == 0) return LMOD_OK;
  hipLaunchKernelGGL(rowloss_fwd_kernel, dim3(R), dim3(256), 0, stream, (const bf16_t*)s, ld_s, Vs,
                     (const bf16_t*)t, ld_t, Va, label, stats);
  return lmod_launch_status();
}

int lmod_rowloss_bwd(const void* s, long long ld_s, int Vs, const void* t, long long ld_t, int Va,
                     const int* label, const float* stats, const float* kd_w, const float* ce_w,
                     const int* seg_id, const float* kd_scale, const float* ce_scale, void* ds, long long ld_ds,
                     int R, hipStream_t stream) {
  if (!s || !stats || !ds || R < 0 || Vs <= 0 || (Vs & 7) || Va < 0 || (Va & 7) || Va > Vs || (ld_s & 7) ||
      (ld_ds & 7) || (t && (ld_t & 7))) return LMOD_EINVAL;
  if (R == 0) return LMOD_OK;
  hipLaunchKernelGGL(rowloss_bwd_kernel, dim3(R), dim3(256), 0, stream, (const bf16_t*)s, ld_s, Vs,
                     (const bf16_t*)t, ld_t, Va, label, stats, kd_w, ce_w, seg_id, kd_scale, ce_scale,
                     (bf16_t*)ds, ld_ds);
  return lmod_launch_status();
}

int lmod_segment_wsum(const float* val, int stride, int col, const float* w, const int* seg_off, int nseg,
                      float* out_sum, float* out_w, hipStream_t stream) {
  if (!val || !seg_off || !out_sum || nseg < 0 || stride <= 0 || col < 0 || col >= stride) return LMOD_EINVAL;
  if (nseg == 0) return LMOD_OK;
  hipLaunchKernelGGL(segment_wsum_kernel, dim3(nseg), dim3(256), 0, stream, val, stride, col, w, seg_off, out_sum, out_w);
  return lmod_launch_status();
}

int lmod_dpo_loss(const float* policy_chosen, const float* policy_rejected, const float* ref_chosen,
                  const float* ref_rejected, int B, float beta, float label_smoothing, int loss_type,
                  float* losses, float* chosen_rewards, float* rejected_rewards, float* d_policy_chosen,
                  float* d_policy_rejected, hipStream_t stream) {
  if (!policy_chosen || !policy_rejected || !ref_chosen || !ref_rejected || !losses || !chosen_rewards ||
      !rejected_rewards || !d_policy_chosen || !d_policy_rejected || B <= 0 || B > 256 || loss_type < 0 ||
      loss_type > 3) return LMOD_EINVAL;
  hipLaunchKernelGGL(dpo_loss_kernel, dim3(1), dim3(256), 0, stream, policy_chosen, policy_rejected, ref_chosen,
                     ref_rejected, B, beta, label_smoothing, loss_type, losses, chosen_rewards, rejected_rewards,
                     d_policy_chosen, d_policy_rejected);
  return lmod_launch_status();
}

int lmod_row_softmax_f32(const float* logits, long long ld, int Va, int log_flag, float* out, int R, hipStream_t stream) {
  if (!logits || !out || R < 0 || Va <= 0 || ld < Va) return LMOD_EINVAL;
  if (R == 0) return LMOD_OK;
  hipLaunchKernelGGL(row_softmax_f32_kernel, dim3(R), dim3(256), 0, stream, logits, ld, Va, log_flag, out);
  return lmod_launch_status();
}

int lmod_rowdot_masked(const float* p, const float* logp, int V, int R, float* x, hipStream_t stream) {
  if (!p || !logp || !x || R < 0 || V <= 0) return LMOD_EINVAL;
  if (R == 0) return LMOD_OK;
  hipLaunchKernelGGL(rowdot_masked_kernel, dim3(R), dim3(256), 0, stream, p, logp, V, x);
  return lmod_launch_status();
}

int lmod_row_argmax_bf16(const void* logits, long long ld, int V, int* out, int R, hipStream_t stream) {
  if (R < 0 || V <= 0 || ld < V) return LMOD_EINVAL;
  if (R == 0) return LMOD_OK;
  if (!logits || !out) return LMOD_EINVAL;
  hipLaunchKernelGGL(row_argmax_kernel, dim3(R), dim3(256), 0, stream, (const bf16_t*)logits, ld, V, out);
  return lmod_launch_status();
}

}  // extern "C"
