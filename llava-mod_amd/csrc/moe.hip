// Sparse-MoE routing kernels (gfx950): a sparse restatement of DeepSpeed-0.9.5's dense one-hot
// MoE (deepspeed.moe.sharded_moe.{TopKGate, top2gating, MOELayer} — third-party, pinned by
// reference requirements.txt:9; constructed at llava_qwen2_moe.py:536-546, result consumed at
// :161-167).  DeepSpeed builds [S,E,C] one-hot dispatch/combine tensors and two einsums; here the
// same decisions (which tokens keep a capacity slot, in token order) are produced as index maps:
//   router_fwd   logits[T,E] = x.float() @ wg^T                (fp32, wave per 4 tokens)
//   gate_top2    softmax, 1st/2nd expert (2nd over logits + noise with the 1st masked)
//   scan         token-order exclusive prefix per expert (cumsum semantics), me/ce sums, l_aux
//   finalize     capacity drop, renormalised combine weights, slot maps
//   combine_fwd/bwd, dispatch_bwd(+router dx), router_wgrad
// Dispatch itself is lmod_gather_rows with slot_token as the index (empty slots -> zero rows).
#include "common.h"

// Experts per layer: the per-token kernels keep one value per expert in registers, so they are compiled for ME = 8 / 16 / 32
// slots and dispatched on E (`ME_DISPATCH`): E <= 8 — the reference shells' 4 and config 5's 8 — runs exactly the code it
// always ran; `--num_experts` up to LMOD_MAX_EXPERTS = 32 takes the wider instantiations.  The two kernels whose per-expert
// state is a register TILE (router logits: 4 tokens x E; router wgrad: E x 8 columns) walk the experts in groups of 8
// (grid.y / grid.z), so their register footprint does not grow with E.
#define MAXE 32
#define EG 8
#define ME_DISPATCH(E, CALL) do { if ((E) <= 8) { constexpr int ME = 8; CALL; } else if ((E) <= 16) { constexpr int ME = 16; CALL; } \
                                 else { constexpr int ME = 32; CALL; } } while (0)
static inline int me_of(int E) { return E <= 8 ? 8 : (E <= 16 ? 16 : 32); }

// ---------------------------------------------------------------- router logits (fp32)
__global__ __launch_bounds__(256) void router_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wg,
                                                        float* __restrict__ logits, int T, int H, int E) {
  const int lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;     // 4 tokens per wave
  if (t0 >= T) return;
  const int eb = blockIdx.y * EG;                               // this block's group of 8 experts
  wg += (long long)eb * H;
  const int En = min(E - eb, EG);
  float acc[4][EG];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < EG; ++e) acc[a][e] = 0.f;
  const int nch = H >> 3;
  for (int c = lane; c < nch; c += 64) {
    float xv[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (t0 + a < T) {
        const u32x4 v = *(const u32x4*)(x + (long long)(t0 + a) * H + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { xv[a][2 * k] = bflo(v[k]); xv[a][2 * k + 1] = bfhi(v[k]); }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[a][k] = 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      if (e < En) {
        const f32x4 w0 = *(const f32x4*)(wg + (long long)e * H + c * 8);
        const f32x4 w1 = *(const f32x4*)(wg + (long long)e * H + c * 8 + 4);
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[a][e] += xv[a][0] * w0[0] + xv[a][1] * w0[1] + xv[a][2] * w0[2] + xv[a][3] * w0[3] +
                       xv[a][4] * w1[0] + xv[a][5] * w1[1] + xv[a][6] * w1[2] + xv[a][7] * w1[3];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      if (e < En) {
        const float s = wave_sum(acc[a][e]);
        if (lane == 0 && t0 + a < T) logits[(long long)(t0 + a) * E + eb + e] = s;
      }
    }
}

// ---------------------------------------------------------------- counter-based random numbers (Philox4x32-10)
// One call = 4 x 32 random bits for (key = seed, counter = (token + offset, stream)): no state, any token can be drawn by any
// thread, reproducible from (seed, offset).  Replaces the torch.rand / log launches in front of every MoE layer
// (sharded_moe.gumbel_rsample draws the top-2 noise; exp_selection_uniform_map the top-1 random-token-selection noise).
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// u in (0, 1): 24 random bits + half a step.  mode 1: Gumbel(0,1) = -log(-log(u)); mode 2: the uniform itself.
template <int ME>
__device__ __forceinline__ void draw_noise(float (&nz)[ME], long long t, int E, unsigned long long seed,
                                           unsigned long long offset, int mode) {
  const unsigned long long ctr = offset + (unsigned long long)t;
#pragma unroll
  for (int grp = 0; grp < ME / 4; ++grp) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)grp, 0u};
    if (grp * 4 < E) philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = (float)(c[j] >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;
      nz[grp * 4 + j] = (mode == 1) ? -logf(-logf(u)) : u;
    }
  }
}

// ---------------------------------------------------------------- per-token softmax + top-2 picks
// noise_mode 0: `noise` ([T,E], may be NULL) is what the caller supplies; 1 / 2: Gumbel / uniform noise is drawn here
// (and written to noise_out if the caller wants to see it).  k == 2: the noise is added to the logits for the SECOND
// pick (top2gating).  k == 1: the noise is the random-token-selection priority (top1gating, use_rts), consumed later.
template <int ME>
__global__ __launch_bounds__(256) void gate_top2_kernel(const float* __restrict__ logits, const float* __restrict__ noise,
                                                       float* __restrict__ gates, int* __restrict__ idx1,
                                                       int* __restrict__ idx2, int T, int E, int k, int noise_mode,
                                                       unsigned long long seed, unsigned long long offset,
                                                       float* __restrict__ noise_out, float* __restrict__ prio) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float nz[ME];
#pragma unroll
  for (int e = 0; e < ME; ++e) nz[e] = 0.f;
  bool have_noise = (noise != nullptr);
  if (noise_mode) {
    draw_noise(nz, t, E, seed, offset, noise_mode);
    have_noise = true;
    if (noise_out) {
#pragma unroll
      for (int e = 0; e < ME; ++e) if (e < E) noise_out[(long long)t * E + e] = nz[e];
    }
  } else if (noise) {
#pragma unroll
    for (int e = 0; e < ME; ++e) if (e < E) nz[e] = noise[(long long)t * E + e];
  }
  float l[ME], g[ME];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < ME; ++e) { l[e] = (e < E) ? logits[(long long)t * E + e] : -INFINITY; mx = fmaxf(mx, l[e]); }
  float z = 0.f;
#pragma unroll
  for (int e = 0; e < ME; ++e) { g[e] = (e < E) ? expf(l[e] - mx) : 0.f; z += g[e]; }
  int i1 = 0; float b1 = -INFINITY;
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    g[e] /= z;
    if (e < E) { gates[(long long)t * E + e] = g[e]; if (g[e] > b1) { b1 = g[e]; i1 = e; } }  // first max wins
  }
  idx1[t] = i1;
  if (k >= 2) {
    int i2 = 0; float b2 = -INFINITY; bool any = false;
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      if (e < E && e != i1) {
        const float v = l[e] + nz[e];
        if (!any || v > b2) { b2 = v; i2 = e; any = true; }
      }
    }
    idx2[t] = i2;
  } else if (prio && have_noise) {          // mask1 * uniform: the priority of this token inside its expert's queue
    float pv = 0.f;
#pragma unroll
    for (int e = 0; e < ME; ++e) if (e == i1) pv = nz[e];
    prio[t] = pv;
  }
}

// ---------------------------------------------------------------- top-1 random token selection (use_rts)
// top1gating keeps, per expert, the C tokens with the LARGEST priority (torch.topk over mask1 * uniform) and then numbers
// the survivors in token order.  One block per expert: bisection on the float bit pattern (priorities are >= 0, so bit
// order = value order) for the smallest threshold that keeps <= C tokens; ties at the boundary are filled in token order.
// keep_idx[t] = idx1[t] if token t keeps its slot, -1 otherwise.
__global__ __launch_bounds__(1024) void rts_select_kernel(const int* __restrict__ idx1, const float* __restrict__ prio,
                                                         int* __restrict__ keep_idx, int T, int C) {
  __shared__ int red[16];
  __shared__ int bcast;
  const int e = blockIdx.x, tid = threadIdx.x;
  auto count_ge = [&](uint32_t thr) {
    int c = 0;
    for (int t = tid; t < T; t += 1024) c += (idx1[t] == e && __float_as_uint(prio[t]) >= thr) ? 1 : 0;
    c = (int)wave_sum((float)c);             // counts <= 2^24: exact in fp32
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    return tot;
  };
  const int total = count_ge(0u);
  uint32_t thr = 0u;
  int kept = total;
  if (total > C) {
    uint32_t lo = 0u, hi = 0x7f800000u;      // f(lo) > C, f(hi) = 0 <= C ; find the smallest thr with f(thr) <= C
    while (hi - lo > 1u) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (count_ge(mid) <= C) hi = mid; else lo = mid;
    }
    thr = hi;
    kept = count_ge(thr);
  }
  int need = (total > C) ? C - kept : 0;     // > 0 only if several tokens tie just below the threshold
  uint32_t tie = 0u;
  if (need > 0) {                            // the tied value = thr - 1 (lo): bisection ended with f(lo) > C >= f(hi)
    tie = thr - 1u;
    if (tid == 0) bcast = 0;
    __syncthreads();
  }
  for (int t = tid; t < T; t += 1024)
    if (idx1[t] == e) keep_idx[t] = (__float_as_uint(prio[t]) >= thr) ? e : -1;     // each token has exactly one owner block
  if (need > 0) {
    __syncthreads();
    if (tid == 0) {                          // rare: first `need` tied tokens in token order
      for (int t = 0; t < T && need > 0; ++t)
        if (idx1[t] == e && __float_as_uint(prio[t]) == tie) { keep_idx[t] = e; --need; }
    }
  }
}

// ---------------------------------------------------------------- token-order prefix (three small launches)
// loc1[t] = #{t' < t : idx1[t'] == idx1[t]} ; loc2[t] = #{t' < t : idx2[t'] == idx2[t]} + count1[idx2[t]]
// (cumsum(mask,0)-1 and the "+ sum(mask1)" offset of top2gating).  Also exp_counts, sum of gates per
// expert (me*T) and l_aux = mean(me*ce)*E*E.
// A single-block scan left the chip idle for ~190 us per MoE layer; now: (1) every 512-token block counts its picks per
// expert and sums its gates, (2) one small block turns the per-block counts into exclusive bases (fixed order, so the
// gate sums / l_aux are deterministic), (3) every block ranks its tokens with wave ballots and adds its base.
// part layout (caller scratch, ints): [NB][2*ME] counts -> bases, then [NB][ME] gate partial sums (as floats).
#define SCAN_BLK 512
template <int ME>
__global__ __launch_bounds__(SCAN_BLK) void moe_count_kernel(const int* __restrict__ idx1, const int* __restrict__ idx2,
                                                            const float* __restrict__ gates, int* __restrict__ part,
                                                            float* __restrict__ gpart, int T, int E, int k) {
  __shared__ int wc[SCAN_BLK / 64][2 * ME];
  __shared__ float wg[SCAN_BLK / 64][ME];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x * SCAN_BLK + tid;
  const int a = (t < T) ? idx1[t] : -1, b = (t < T && k >= 2) ? idx2[t] : -1;
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    const int c1 = __popcll(__ballot(a == e)), c2 = __popcll(__ballot(b == e));
    const float gs = wave_sum((t < T && e < E) ? gates[(long long)t * E + e] : 0.f);
    if (lane == 0) { wc[w][e] = c1; wc[w][ME + e] = c2; wg[w][e] = gs; }
  }
  __syncthreads();
  if (tid < 2 * ME) {
    int s = 0;
#pragma unroll
    for (int x = 0; x < SCAN_BLK / 64; ++x) s += wc[x][tid];
    part[blockIdx.x * 2 * ME + tid] = s;
  } else if (tid >= 64 && tid < 64 + ME) {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < SCAN_BLK / 64; ++x) s += wg[x][tid - 64];
    gpart[blockIdx.x * ME + tid - 64] = s;
  }
}

template <int ME>
__global__ __launch_bounds__(128) void moe_bases_kernel(int* __restrict__ part, const float* __restrict__ gpart, int nb,
                                                      int* __restrict__ exp_counts, float* __restrict__ gate_sum,
                                                      float* __restrict__ l_aux, int* __restrict__ slots_used, int T, int E,
                                                      int k, int C) {
  __shared__ int tot[2 * ME];
  __shared__ float gsum[ME];
  const int tid = threadIdx.x;
  if (tid < 2 * ME) {                      // thread = (pick, expert): exclusive prefix over the blocks, in block order
    int run = 0;
    for (int x = 0; x < nb; ++x) { const int c = part[x * 2 * ME + tid]; part[x * 2 * ME + tid] = run; run += c; }
    tot[tid] = run;
  } else if (tid >= 64 && tid < 64 + ME) {
    float s = 0.f;
    for (int x = 0; x < nb; ++x) s += gpart[x * ME + tid - 64];
    gsum[tid - 64] = s;
  }
  __syncthreads();
  if (tid >= ME && tid < 2 * ME) {       // second picks queue behind ALL first picks of the same expert
    const int off = tot[tid - ME];
    for (int x = 0; x < nb; ++x) part[x * 2 * ME + tid] += off;
  }
  if (tid == 0 && exp_counts) {
    float la = 0.f;
    for (int e = 0; e < E; ++e) {
      gate_sum[e] = gsum[e];
      exp_counts[e] = tot[e];
      // capacity slots are filled densely from 0: first picks, then second picks behind them
      slots_used[e] = min(C, tot[e] + ((k >= 2) ? tot[ME + e] : 0));
      la += (gsum[e] / (float)T) * ((float)tot[e] / (float)T);
    }
    // top2gating: mean(me*ce)*E*E ; top1gating: sum(me*ce)*E — both equal E * sum_e(me*ce)
    l_aux[0] = la * (float)E;
  }
}

template <int ME>
__global__ __launch_bounds__(SCAN_BLK) void moe_rank_kernel(const int* __restrict__ idx1, const int* __restrict__ idx2,
                                                           const int* __restrict__ part, int* __restrict__ loc1,
                                                           int* __restrict__ loc2, int T, int k) {
  __shared__ int wc[SCAN_BLK / 64][2 * ME];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x * SCAN_BLK + tid;
  const int a = (t < T) ? idx1[t] : -1, b = (t < T && k >= 2) ? idx2[t] : -1;
  const unsigned long long below = (1ull << lane) - 1ull;
  int r1 = 0, r2 = 0;
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    const unsigned long long m1 = __ballot(a == e), m2 = __ballot(b == e);
    if (a == e) r1 = __popcll(m1 & below);
    if (b == e) r2 = __popcll(m2 & below);
    if (lane == 0) { wc[w][e] = __popcll(m1); wc[w][ME + e] = __popcll(m2); }
  }
  __syncthreads();
  if (t >= T) return;
  const int* base = part + blockIdx.x * 2 * ME;
  if (a < 0) { loc1[t] = 0x7fffffff; return; }      // top-1 random token selection dropped this token (k == 1 only)
  int p1 = base[a] + r1;
  for (int x = 0; x < w; ++x) p1 += wc[x][a];
  loc1[t] = p1;
  if (k >= 2) {
    int p2 = base[ME + b] + r2;
    for (int x = 0; x < w; ++x) p2 += wc[x][ME + b];
    loc2[t] = p2;
  }
}

// ---------------------------------------------------------------- capacity drop + combine weights + slot maps
__global__ __launch_bounds__(256) void moe_finalize_kernel(const float* __restrict__ gates, const int* __restrict__ idx1,
                                                          const int* __restrict__ idx2, const int* __restrict__ loc1,
                                                          const int* __restrict__ loc2, int* __restrict__ slot1,
                                                          int* __restrict__ slot2, float* __restrict__ w1,
                                                          float* __restrict__ w2, int* __restrict__ slot_token,
                                                          float* __restrict__ slot_w, int T, int E, int C, int k) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int e1 = idx1[t], l1 = loc1[t];
  const bool k1 = l1 < C;
  float g1 = k1 ? gates[(long long)t * E + e1] : 0.f;
  float g2 = 0.f; int e2 = 0, l2 = 0; bool k2 = false;
  if (k >= 2) { e2 = idx2[t]; l2 = loc2[t]; k2 = l2 < C; g2 = k2 ? gates[(long long)t * E + e2] : 0.f; }
  float a1 = g1, a2 = g2;
  if (k >= 2) {                                   // top-2 renormalises over the surviving picks
    const float den = fmaxf(g1 + g2, 1.1920929e-07f);
    a1 = g1 / den; a2 = g2 / den;
  }
  const int s1 = k1 ? e1 * C + l1 : -1;
  const int s2 = k2 ? e2 * C + l2 : -1;
  slot1[t] = s1; w1[t] = a1;
  if (k >= 2) { slot2[t] = s2; w2[t] = a2; }
  if (k1) { slot_token[s1] = t; slot_w[s1] = a1; }
  if (k2) { slot_token[s2] = t; slot_w[s2] = a2; }
}

// ---------------------------------------------------------------- combine: out[t] = bf(w1)*y[s1] + bf(w2)*y[s2]
// (combine_weights.type_as(x) in MOELayer.forward -> the weights are rounded to bf16 first)
__global__ __launch_bounds__(256) void moe_combine_fwd_kernel(const bf16_t* __restrict__ y, const int* __restrict__ slot1,
                                                             const int* __restrict__ slot2, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, bf16_t* __restrict__ out,
                                                             long long T, int H) {
  const int nch = H >> 3;
  const long long total = T * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long t = id / nch; const int c = (int)(id - t * nch) * 8;
    const int s1 = slot1[t], s2 = slot2 ? slot2[t] : -1;
    const float a1 = (s1 >= 0) ? bfround(w1[t]) : 0.f, a2 = (s2 >= 0) ? bfround(w2[t]) : 0.f;
    u32x4 v1 = {0u, 0u, 0u, 0u}, v2 = {0u, 0u, 0u, 0u};
    if (s1 >= 0) v1 = *(const u32x4*)(y + (long long)s1 * H + c);
    if (s2 >= 0) v2 = *(const u32x4*)(y + (long long)s2 * H + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack2bf(a1 * bflo(v1[k]) + a2 * bflo(v2[k]), a1 * bfhi(v1[k]) + a2 * bfhi(v2[k]));
    *(u32x4*)(out + t * H + c) = o;
  }
}

// combine backward, slot side: dy[slot] = bf(slot_w) * dout[slot_token]  (zero rows for empty slots)
__global__ __launch_bounds__(256) void moe_combine_bwd_slots_kernel(const bf16_t* __restrict__ dout,
                                                                   const int* __restrict__ slot_token,
                                                                   const float* __restrict__ slot_w,
                                                                   bf16_t* __restrict__ dy, long long S, int H) {
  const int nch = H >> 3;
  const long long total = S * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long s = id / nch; const int c = (int)(id - s * nch) * 8;
    const int t = slot_token[s];
    u32x4 o = {0u, 0u, 0u, 0u};
    if (t >= 0) {
      const float a = bfround(slot_w[s]);
      const u32x4 v = *(const u32x4*)(dout + (long long)t * H + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pack2bf(a * bflo(v[k]), a * bfhi(v[k]));
    }
    *(u32x4*)(dy + s * H + c) = o;
  }
}

// combine backward, weight side: dw_j[t] = <dout[t], y[slot_j(t)]>   (wave per token)
__global__ __launch_bounds__(256) void moe_combine_bwd_w_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                               const int* __restrict__ slot1, const int* __restrict__ slot2,
                                                               float* __restrict__ dw1, float* __restrict__ dw2, int T, int H) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int s1 = slot1[t], s2 = slot2 ? slot2[t] : -1;
  float a1 = 0.f, a2 = 0.f;
  const int nch = H >> 3;
  for (int c = lane; c < nch; c += 64) {
    const u32x4 d = *(const u32x4*)(dout + (long long)t * H + c * 8);
    if (s1 >= 0) {
      const u32x4 v = *(const u32x4*)(y + (long long)s1 * H + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) a1 += bflo(d[k]) * bflo(v[k]) + bfhi(d[k]) * bfhi(v[k]);
    }
    if (s2 >= 0) {
      const u32x4 v = *(const u32x4*)(y + (long long)s2 * H + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) a2 += bflo(d[k]) * bflo(v[k]) + bfhi(d[k]) * bfhi(v[k]);
    }
  }
  a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) { dw1[t] = a1; if (dw2) dw2[t] = a2; }
}

// gate backward: (dw1, dw2, d l_aux) -> dlogits[T,E]
template <int ME>
__global__ __launch_bounds__(256) void moe_gate_bwd_kernel(const float* __restrict__ gates, const int* __restrict__ idx1,
                                                          const int* __restrict__ idx2, const int* __restrict__ slot1,
                                                          const int* __restrict__ slot2, const float* __restrict__ dw1,
                                                          const float* __restrict__ dw2, const int* __restrict__ exp_counts,
                                                          const float* __restrict__ d_laux, float* __restrict__ dlogits,
                                                          int T, int E, int k) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float g[ME], dg[ME];
  const float dla = d_laux ? d_laux[0] : 0.f;
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    g[e] = (e < E) ? gates[(long long)t * E + e] : 0.f;
    // l_aux = E * sum_e (sum_t gates[t,e]/T) * (count1[e]/T)
    dg[e] = (e < E) ? dla * (float)E * ((float)exp_counts[e] / (float)T) / (float)T : 0.f;
  }
  const int e1 = idx1[t];
  const bool k1 = slot1[t] >= 0;
  if (k >= 2) {
    const int e2 = idx2[t];
    const bool k2 = slot2[t] >= 0;
    const float g1 = k1 ? g[e1] : 0.f, g2 = k2 ? g[e2] : 0.f;
    const float sum = g1 + g2;
    if (sum > 1.1920929e-07f) {
      const float u1 = dw1[t], u2 = dw2[t];
      const float common = (u1 * g1 + u2 * g2) / (sum * sum);
      const float d1 = u1 / sum - common, d2 = u2 / sum - common;
#pragma unroll
      for (int e = 0; e < ME; ++e) { if (k1 && e == e1) dg[e] += d1; if (k2 && e == e2) dg[e] += d2; }
    }
  } else {
    const float u1 = dw1[t];
#pragma unroll
    for (int e = 0; e < ME; ++e) if (k1 && e == e1) dg[e] += u1;
  }
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < ME; ++e) dot += g[e] * dg[e];
#pragma unroll
  for (int e = 0; e < ME; ++e) if (e < E) dlogits[(long long)t * E + e] = g[e] * (dg[e] - dot);
}

// dx[t] = d_in[slot1(t)] + d_in[slot2(t)] + bf16( sum_e dlogits[t,e] * wg[e] )   (dispatch is a copy)
template <int ME>
__global__ __launch_bounds__(256) void moe_dispatch_bwd_kernel(const bf16_t* __restrict__ d_in, const int* __restrict__ slot1,
                                                              const int* __restrict__ slot2, const float* __restrict__ dlogits,
                                                              const float* __restrict__ wg, bf16_t* __restrict__ dx,
                                                              long long T, int H, int E) {
  const int nch = H >> 3;
  const long long total = T * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long t = id / nch; const int c = (int)(id - t * nch) * 8;
    const int s1 = slot1[t], s2 = slot2 ? slot2[t] : -1;
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = 0.f;
    if (dlogits) {
#pragma unroll
      for (int e = 0; e < ME; ++e) {
        if (e < E) {
          const float dl = dlogits[t * E + e];
          const f32x4 a = *(const f32x4*)(wg + (long long)e * H + c);
          const f32x4 b = *(const f32x4*)(wg + (long long)e * H + c + 4);
          r[0] += dl * a[0]; r[1] += dl * a[1]; r[2] += dl * a[2]; r[3] += dl * a[3];
          r[4] += dl * b[0]; r[5] += dl * b[1]; r[6] += dl * b[2]; r[7] += dl * b[3];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = bfround(r[k]);
    }
    if (s1 >= 0) {
      const u32x4 v = *(const u32x4*)(d_in + (long long)s1 * H + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) { r[2 * k] += bflo(v[k]); r[2 * k + 1] += bfhi(v[k]); }
    }
    if (s2 >= 0) {
      const u32x4 v = *(const u32x4*)(d_in + (long long)s2 * H + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) { r[2 * k] += bflo(v[k]); r[2 * k + 1] += bfhi(v[k]); }
    }
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(r[2 * k], r[2 * k + 1]);
    *(u32x4*)(dx + t * H + c) = o;
  }
}

// router wgrad: partial[slab][e][h] = sum_{t in slab} dlogits[t,e] * x[t,h] ; then reduced into dwg (+=).
// One thread owns 8 consecutive h (16-byte loads of x); a block covers 256*8 columns x WG_SLAB tokens, the slab's
// dlogits rows sit in LDS.  (The first version read x two bytes at a time: 167 us for 134 MB.)
#define WG_SLAB 64
__global__ __launch_bounds__(256) void router_wgrad_partial_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dlogits,
                                                                  float* __restrict__ partial, int T, int H, int E) {
  __shared__ float dl[WG_SLAB * EG];
  const int h0 = (blockIdx.x * 256 + threadIdx.x) * 8;
  const int slab = blockIdx.y;
  const int eb = blockIdx.z * EG, En = min(E - eb, EG);         // this block's group of 8 experts
  const int lo = slab * WG_SLAB, hi = min(lo + WG_SLAB, T);
  for (int i = threadIdx.x; i < (hi - lo) * En; i += 256) dl[(i / En) * EG + (i % En)] = dlogits[(long long)(lo + i / En) * E + eb + (i % En)];
  __syncthreads();
  if (h0 >= H) return;
  float acc[EG][8];
#pragma unroll
  for (int e = 0; e < EG; ++e)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[e][j] = 0.f;
#pragma unroll 4
  for (int t = lo; t < hi; ++t) {
    const u32x4 xv = *(const u32x4*)(x + (long long)t * H + h0);
    float xf[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) { xf[2 * w] = bflo(xv[w]); xf[2 * w + 1] = bfhi(xv[w]); }
#pragma unroll
    for (int e = 0; e < EG; ++e)
      if (e < En) {
        const float d = dl[(t - lo) * EG + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[e][j] += d * xf[j];
      }
  }
#pragma unroll
  for (int e = 0; e < EG; ++e)
    if (e < En) {
      float* pp = partial + ((long long)slab * E + eb + e) * H + h0;
      *(f32x4*)pp = (f32x4){acc[e][0], acc[e][1], acc[e][2], acc[e][3]};
      *(f32x4*)(pp + 4) = (f32x4){acc[e][4], acc[e][5], acc[e][6], acc[e][7]};
    }
}
// 64 outputs per block x 16 slab groups: every thread sums its group's partials, the 16 group sums are added in fixed order
__global__ __launch_bounds__(1024) void router_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dwg,
                                                                  int nslab, int EH, int accumulate) {
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < EH)
    for (int b = grp; b < nslab; b += 16) s += partial[(long long)b * EH + i];
  red[grp][o] = s;
  __syncthreads();
  if (grp == 0 && i < EH) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][o];
    dwg[i] = accumulate ? dwg[i] + t : t;
  }
}

static inline int grid_for(long long work, int cap = 256 * 16) {
  long long b = (work + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ---------------------------------------------------------------- Residual-MoE mix (deepspeed.moe.layer.MoE, use_residual)
//   coef = softmax(Linear(hidden, 2)(x));  out = moe_out * coef[..., 0:1] + mlp(x) * coef[..., 1:]
// One wave per token row.  Roundings follow the bf16 module: the 2 coefficient logits (fp32 dot + bias) are rounded to
// bf16, so is the softmax output, each product and the sum.  p[T,2] (fp32, the bf16-rounded coefficients) is kept for backward.
__global__ __launch_bounds__(256) void residual_mix_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                              const float* __restrict__ clog, const float* __restrict__ cbias,
                                                              bf16_t* __restrict__ out, float* __restrict__ p, int T, int H) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float c0 = bfround(clog[t * 2] + cbias[0]), c1 = bfround(clog[t * 2 + 1] + cbias[1]);
  const float m = fmaxf(c0, c1), e0 = expf(c0 - m), e1 = expf(c1 - m);
  const float p0 = bfround(e0 / (e0 + e1)), p1 = bfround(e1 / (e0 + e1));
  if (lane == 0) { p[t * 2] = p0; p[t * 2 + 1] = p1; }
  for (int c = lane * 8; c < H; c += 512) {
    const u32x4 av = *(const u32x4*)(a + t * H + c), bv = *(const u32x4*)(b + t * H + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = bfround(bflo(av[k]) * p0) + bfround(bflo(bv[k]) * p1);
      const float hi = bfround(bfhi(av[k]) * p0) + bfround(bfhi(bv[k]) * p1);
      o[k] = pack2bf(lo, hi);
    }
    *(u32x4*)(out + t * H + c) = o;
  }
}
// d_a = dout * p0, d_b = dout * p1 (bf16); dc[t] = softmax backward of (dp0, dp1) = (sum_h dout*a, sum_h dout*b)
__global__ __launch_bounds__(256) void residual_mix_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ a,
                                                              const bf16_t* __restrict__ b, const float* __restrict__ p,
                                                              bf16_t* __restrict__ da, bf16_t* __restrict__ db,
                                                              float* __restrict__ dc, int T, int H) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float p0 = p[t * 2], p1 = p[t * 2 + 1];
  float s0 = 0.f, s1 = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    const u32x4 dv = *(const u32x4*)(dout + t * H + c), av = *(const u32x4*)(a + t * H + c), bv = *(const u32x4*)(b + t * H + c);
    u32x4 oa, ob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dl = bflo(dv[k]), dh = bfhi(dv[k]);
      s0 += dl * bflo(av[k]) + dh * bfhi(av[k]);
      s1 += dl * bflo(bv[k]) + dh * bfhi(bv[k]);
      oa[k] = pack2bf(dl * p0, dh * p0);
      ob[k] = pack2bf(dl * p1, dh * p1);
    }
    *(u32x4*)(da + t * H + c) = oa;
    *(u32x4*)(db + t * H + c) = ob;
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if (lane == 0) {
    const float dot = p0 * s0 + p1 * s1;
    dc[t * 2] = p0 * (s0 - dot);
    dc[t * 2 + 1] = p1 * (s1 - dot);
  }
}
// dx[t, :] = sum_e dlogits[t, e] * w[e, :]   (the input gradient of a tiny fp32 linear: router-sized heads)
template <int ME>
__global__ __launch_bounds__(256) void small_linear_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                bf16_t* __restrict__ dx, int T, int H, int E) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float d[ME];
#pragma unroll
  for (int e = 0; e < ME; ++e) d[e] = (e < E) ? dl[t * E + e] : 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      if (e < E) {
        const f32x4 w0 = *(const f32x4*)(w + (long long)e * H + c), w1 = *(const f32x4*)(w + (long long)e * H + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += d[e] * w0[j]; acc[4 + j] += d[e] * w1[j]; }
      }
    }
    *(u32x4*)(dx + t * H + c) = (u32x4){pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])};
  }
}

extern "C" {

int lmod_moe_router_fwd(const void* x, const float* wg, float* logits, int T, int H, int E, hipStream_t stream) {
  if (!x || !wg || !logits || T < 0 || H <= 0 || (H & 7) || E <= 0 || E > MAXE) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(router_fwd_kernel, dim3((T + 15) / 16, (E + EG - 1) / EG), dim3(256), 0, stream, (const bf16_t*)x, wg, logits, T, H, E);
  return lmod_launch_status();
}

// Full gating decision from logits.
// noise / noise_mode: k == 2: additive [T,E] noise for the 2nd pick (Gumbel in the reference) — supplied (mode 0, may be
//   NULL = no noise) or drawn in the kernel from (seed, offset) (mode 1).  k == 1: random-token-selection priorities
//   (uniform in the reference, use_rts=True): supplied [T,E] (mode 0; NULL = token order, use_rts=False) or drawn (mode 2).
//   noise_out (nullable, [T,E]): receives the drawn noise.
// Outputs: gates[T,E] f32; idx1/idx2/slot1/slot2 [T] i32; w1/w2 [T] f32; slot_token [E*C] i32 (-1 empty);
// slot_w [E*C] f32; exp_counts [E] i32; gate_sum [E] f32; l_aux [1] f32.
// scratch: 4*T + 3*ME*ceil(T/512) i32, ME = 8 / 16 / 32 expert slots for E <= 8 / 16 / 32 (loc1, loc2, rts priorities / kept
// picks, per-block pick counts / bases, gate partials).
int lmod_moe_gate(const float* logits, const float* noise, int T, int E, int k, int C, float* gates, int* idx1,
                  int* idx2, int* slot1, int* slot2, float* w1, float* w2, int* slot_token, float* slot_w,
                  int* exp_counts, float* gate_sum, float* l_aux, int* slots_used, int* scratch, int noise_mode,
                  unsigned long long seed, unsigned long long offset, float* noise_out, hipStream_t stream) {
  if (!logits || !gates || !idx1 || !slot1 || !w1 || !slot_token || !slot_w || !exp_counts || !gate_sum || !l_aux ||
      !slots_used || !scratch || T <= 0 || E <= 0 || E > MAXE || (k != 1 && k != 2) || C <= 0) return LMOD_EINVAL;
  if (k == 2 && (!idx2 || !slot2 || !w2)) return LMOD_EINVAL;
  if (noise_mode < 0 || noise_mode > 2 || (noise_mode == 1 && k != 2) || (noise_mode == 2 && k != 1)) return LMOD_EINVAL;
  int* loc1 = scratch; int* loc2 = scratch + T;
  float* prio = (float*)(scratch + 2 * (long long)T);
  int* keep = scratch + 3 * (long long)T;
  const int nb = (T + SCAN_BLK - 1) / SCAN_BLK;
  int* part = scratch + 4 * (long long)T;                       // [nb][2*ME] ints
  float* gpart = (float*)(part + (long long)nb * 2 * me_of(E)); // [nb][ME] floats
  const bool rts = (k == 1) && (noise != nullptr || noise_mode == 2);
  const int* rank_idx = idx1;
  ME_DISPATCH(E, {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gate_top2_kernel<ME>), dim3((T + 255) / 256), dim3(256), 0, stream, logits, noise, gates, idx1, idx2,
                       T, E, k, noise_mode, seed, offset, noise_out, rts ? prio : (float*)nullptr);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_count_kernel<ME>), dim3(nb), dim3(SCAN_BLK), 0, stream, idx1, idx2, gates, part, gpart, T, E, k);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_bases_kernel<ME>), dim3(1), dim3(128), 0, stream, part, gpart, nb, exp_counts, gate_sum, l_aux,
                       slots_used, T, E, k, C);
    if (rts) {     // survivors of the random token selection, then their token-order positions (statistics stay as above)
      hipLaunchKernelGGL(rts_select_kernel, dim3(E), dim3(1024), 0, stream, idx1, prio, keep, T, C);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_count_kernel<ME>), dim3(nb), dim3(SCAN_BLK), 0, stream, keep, idx2, gates, part, gpart, T, E, k);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_bases_kernel<ME>), dim3(1), dim3(128), 0, stream, part, gpart, nb, (int*)nullptr,
                         (float*)nullptr, (float*)nullptr, (int*)nullptr, T, E, k, C);
      rank_idx = keep;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_rank_kernel<ME>), dim3(nb), dim3(SCAN_BLK), 0, stream, rank_idx, idx2, part, loc1, loc2, T, k);
  });
  if (hipMemsetAsync(slot_token, 0xFF, (size_t)E * C * 4, stream) != hipSuccess) return LMOD_ELAUNCH;
  if (hipMemsetAsync(slot_w, 0, (size_t)E * C * 4, stream) != hipSuccess) return LMOD_ELAUNCH;
  hipLaunchKernelGGL(moe_finalize_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, gates, idx1, idx2, loc1, loc2,
                     slot1, slot2, w1, w2, slot_token, slot_w, T, E, C, k);
  return lmod_launch_status();
}

int lmod_moe_combine_fwd(const void* y, const int* slot1, const int* slot2, const float* w1, const float* w2,
                         void* out, int T, int H, hipStream_t stream) {
  if (!y || !slot1 || !w1 || !out || T < 0 || H <= 0 || (H & 7) || (slot2 && !w2)) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(moe_combine_fwd_kernel, dim3(grid_for((long long)T * (H >> 3))), dim3(256), 0, stream,
                     (const bf16_t*)y, slot1, slot2, w1, w2, (bf16_t*)out, (long long)T, H);
  return lmod_launch_status();
}

int lmod_moe_combine_bwd(const void* dout, const void* y, const int* slot1, const int* slot2, const int* slot_token,
                         const float* slot_w, void* dy, float* dw1, float* dw2, int T, int S, int H,
                         hipStream_t stream) {
  if (!dout || !y || !slot1 || !slot_token || !slot_w || !dy || !dw1 || T < 0 || S < 0 || H <= 0 || (H & 7))
    return LMOD_EINVAL;
  if (T == 0 || S == 0) return LMOD_OK;
  hipLaunchKernelGGL(moe_combine_bwd_slots_kernel, dim3(grid_for((long long)S * (H >> 3))), dim3(256), 0, stream,
                     (const bf16_t*)dout, slot_token, slot_w, (bf16_t*)dy, (long long)S, H);
  hipLaunchKernelGGL(moe_combine_bwd_w_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)dout,
                     (const bf16_t*)y, slot1, slot2, dw1, dw2, T, H);
  return lmod_launch_status();
}

int lmod_moe_gate_bwd(const float* gates, const int* idx1, const int* idx2, const int* slot1, const int* slot2,
                      const float* dw1, const float* dw2, const int* exp_counts, const float* d_laux, float* dlogits,
                      int T, int E, int k, hipStream_t stream) {
  if (!gates || !idx1 || !slot1 || !dw1 || !exp_counts || !dlogits || T < 0 || E <= 0 || E > MAXE || (k != 1 && k != 2))
    return LMOD_EINVAL;
  if (k == 2 && (!idx2 || !slot2 || !dw2)) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  ME_DISPATCH(E, hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_gate_bwd_kernel<ME>), dim3((T + 255) / 256), dim3(256), 0, stream, gates, idx1, idx2,
                                    slot1, slot2, dw1, dw2, exp_counts, d_laux, dlogits, T, E, k));
  return lmod_launch_status();
}

int lmod_moe_dispatch_bwd(const void* d_in, const int* slot1, const int* slot2, const float* dlogits, const float* wg,
                          void* dx, int T, int H, int E, hipStream_t stream) {
  if (!d_in || !slot1 || !dx || T < 0 || H <= 0 || (H & 7) || (dlogits && (!wg || E <= 0 || E > MAXE))) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  ME_DISPATCH(dlogits ? E : 1, hipLaunchKernelGGL(HIP_KERNEL_NAME(moe_dispatch_bwd_kernel<ME>), dim3(grid_for((long long)T * (H >> 3))), dim3(256), 0,
                                                  stream, (const bf16_t*)d_in, slot1, slot2, dlogits, wg, (bf16_t*)dx, (long long)T, H, E));
  return lmod_launch_status();
}

// workspace: ceil(T/256) * E * H floats
int lmod_moe_router_wgrad(const void* x, const float* dlogits, float* dwg, float* workspace, int T, int H, int E,
                          int accumulate, hipStream_t stream) {
  if (!x || !dlogits || !dwg || !workspace || T < 0 || H <= 0 || (H & 7) || E <= 0 || E > MAXE) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  const int nslab = (T + WG_SLAB - 1) / WG_SLAB;
  hipLaunchKernelGGL(router_wgrad_partial_kernel, dim3((H / 8 + 255) / 256, nslab, (E + EG - 1) / EG), dim3(256), 0, stream,
                     (const bf16_t*)x, dlogits, workspace, T, H, E);
  hipLaunchKernelGGL(router_wgrad_reduce_kernel, dim3((E * H + 63) / 64), dim3(1024), 0, stream, workspace, dwg, nslab,
                     E * H, accumulate);
  return lmod_launch_status();
}

int lmod_moe_residual_mix_fwd(const void* moe_out, const void* mlp_out, const float* coef_logits, const float* coef_bias,
                              void* out, float* p, int T, int H, hipStream_t stream) {
  if (T < 0 || H <= 0 || (H & 7)) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  if (!moe_out || !mlp_out || !coef_logits || !coef_bias || !out || !p) return LMOD_EINVAL;
  hipLaunchKernelGGL(residual_mix_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)moe_out,
                     (const bf16_t*)mlp_out, coef_logits, coef_bias, (bf16_t*)out, p, T, H);
  return lmod_launch_status();
}

int lmod_moe_residual_mix_bwd(const void* dout, const void* moe_out, const void* mlp_out, const float* p, void* d_moe,
                              void* d_mlp, float* d_coef_logits, int T, int H, hipStream_t stream) {
  if (T < 0 || H <= 0 || (H & 7)) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  if (!dout || !moe_out || !mlp_out || !p || !d_moe || !d_mlp || !d_coef_logits) return LMOD_EINVAL;
  hipLaunchKernelGGL(residual_mix_bwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)dout,
                     (const bf16_t*)moe_out, (const bf16_t*)mlp_out, p, (bf16_t*)d_moe, (bf16_t*)d_mlp, d_coef_logits, T, H);
  return lmod_launch_status();
}

int lmod_small_linear_dgrad(const float* dlogits, const float* w, void* dx, int T, int H, int E, hipStream_t stream) {
  if (T < 0 || H <= 0 || (H & 7) || E <= 0 || E > MAXE) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  if (!dlogits || !w || !dx || ((uintptr_t)w & 15)) return LMOD_EINVAL;
  ME_DISPATCH(E, hipLaunchKernelGGL(HIP_KERNEL_NAME(small_linear_dgrad_kernel<ME>), dim3((T + 3) / 4), dim3(256), 0, stream, dlogits, w,
                                    (bf16_t*)dx, T, H, E));
  return lmod_launch_status();
}

}  // extern "C"
