// HBM-bound row kernels of the distillation step (gfx950): RMSNorm fwd/bwd (+ fused residual
// add), LayerNorm fwd (frozen ViT), RoPE fwd/bwd, SwiGLU fwd/bwd, exact-GELU fwd/bwd, residual
// add, row gather (embedding lookup + multimodal splice), im2col for the 14x14/14 patch conv,
// ViT embedding assembly, fused AdamW.  All bf16 traffic moves as 16-byte vectors, one wave per
// row for the reductions (wave-level shuffles only, no LDS).
//
// Rounding order follows the reference's bf16 execution so results track it to bf16 ulp:
//   RMSNorm  qwen2/modeling_qwen2.py:92-97   (fp32 stats, cast to bf16, THEN * weight)
//   RoPE     qwen2/modeling_qwen2.py:138-171 (bf16 tables; each product and the sum round)
//   SwiGLU   qwen2/modeling_qwen2.py:186-187 (silu rounds, product rounds)
#include "common.h"

#define MAXCH 16   // 16-byte chunks cached per lane -> H <= 64*16*8 = 8192

// ------------------------------------------------------------------ RMSNorm
// h = res ? bf16(x + res) : x ;  y = w * bf16(h * rsqrt(mean(h^2) + eps))
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                         const bf16_t* __restrict__ w, bf16_t* __restrict__ hout,
                                                         bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                         int T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nch = H >> 3;
  const bf16_t* xr = x + (long long)row * H;
  u32x4 buf[MAXCH];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      u32x4 v = *(const u32x4*)(xr + c * 8);
      if (res) {
        const u32x4 rv = *(const u32x4*)(res + (long long)row * H + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = pack2bf(bflo(v[k]) + bflo(rv[k]), bfhi(v[k]) + bfhi(rv[k]));
        if (hout) *(u32x4*)(hout + (long long)row * H + c * 8) = v;
      }
      buf[i] = v;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float a = bflo(v[k]), b = bfhi(v[k]); ss += a * a + b * b; }
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const u32x4 v = buf[i];
      const u32x4 wv = *(const u32x4*)(w + c * 8);
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = pack2bf(bflo(wv[k]) * bfround(bflo(v[k]) * rstd), bfhi(wv[k]) * bfround(bfhi(v[k]) * rstd));
      *(u32x4*)(y + (long long)row * H + c * 8) = o;
    }
  }
}

// dh = rstd * (g - xhat * mean(g * xhat)) + dres,  g = dy * w, xhat = h * rstd
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h,
                                                         const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                         const bf16_t* __restrict__ dres, bf16_t* __restrict__ dh,
                                                         int T, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nch = H >> 3;
  const float rstd = rstd_in[row];
  u32x4 gb[MAXCH], hb[MAXCH];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const u32x4 dv = *(const u32x4*)(dy + (long long)row * H + c * 8);
      const u32x4 hv = *(const u32x4*)(h + (long long)row * H + c * 8);
      const u32x4 wv = *(const u32x4*)(w + c * 8);
      u32x4 g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g0 = bflo(dv[k]) * bflo(wv[k]), g1 = bfhi(dv[k]) * bfhi(wv[k]);
        dot += g0 * bflo(hv[k]) + g1 * bfhi(hv[k]);
        g[k] = pack2bf(g0, g1);          // keep g in bf16 pairs to halve register use
      }
      gb[i] = g; hb[i] = hv;
    }
  }
  dot = wave_sum(dot) * rstd / (float)H;   // mean(g * xhat)
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      u32x4 o;
      u32x4 rv = {0u, 0u, 0u, 0u};
      if (dres) rv = *(const u32x4*)(dres + (long long)row * H + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = rstd * (bflo(gb[i][k]) - bflo(hb[i][k]) * rstd * dot) + bflo(rv[k]);
        const float b = rstd * (bfhi(gb[i][k]) - bfhi(hb[i][k]) * rstd * dot) + bfhi(rv[k]);
        o[k] = pack2bf(a, b);
      }
      *(u32x4*)(dh + (long long)row * H + c * 8) = o;
    }
  }
}

// ------------------------------------------------------------------ LayerNorm (ViT, forward only)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                           int T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nch = H >> 3;
  u32x4 buf[MAXCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      buf[i] = *(const u32x4*)(x + (long long)row * H + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) s += bflo(buf[i][k]) + bfhi(buf[i][k]);
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = bflo(buf[i][k]) - mean, bb = bfhi(buf[i][k]) - mean;
        ss += a * a + bb * bb;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const u32x4 wv = *(const u32x4*)(w + c * 8);
      const u32x4 bv = *(const u32x4*)(b + c * 8);
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = pack2bf((bflo(buf[i][k]) - mean) * rstd * bflo(wv[k]) + bflo(bv[k]),
                       (bfhi(buf[i][k]) - mean) * rstd * bfhi(wv[k]) + bfhi(bv[k]));
      *(u32x4*)(y + (long long)row * H + c * 8) = o;
    }
  }
}

// ------------------------------------------------------------------ RoPE (rotate_half form), in place
// buffer [T, ld]; heads 0..nheads-1 of width hd start at column 0 (q heads then k heads of a fused
// QKV row).  cos/sin: [max_pos, hd] bf16 tables (emb = cat(freqs, freqs)).  pos: int32 [T].
// bwd != 0 applies the transpose (gradient) map.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ buf, const bf16_t* __restrict__ cosT,
                                                  const bf16_t* __restrict__ sinT, const int* __restrict__ pos,
                                                  int T, int nheads, int hd, int ld, int bwd) {
  const int per_head = hd >> 4;                    // threads per head: each does 8 (i) + 8 (i+half)
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)T * nheads * per_head;
  if (gid >= total) return;
  const int j = (int)(gid % per_head);
  const long long th = gid / per_head;
  const int head = (int)(th % nheads);
  const long long t = th / nheads;
  const int half = hd >> 1;
  bf16_t* p = buf + t * ld + head * hd + j * 8;
  const int ps = pos[t];
  const bf16_t* cp = cosT + (long long)ps * hd + j * 8;
  const bf16_t* sp = sinT + (long long)ps * hd + j * 8;
  const u32x4 x1 = *(const u32x4*)p, x2 = *(const u32x4*)(p + half);
  const u32x4 c1 = *(const u32x4*)cp, c2 = *(const u32x4*)(cp + half);
  const u32x4 s1 = *(const u32x4*)sp, s2 = *(const u32x4*)(sp + half);
  u32x4 o1, o2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a0 = bflo(x1[k]), a1 = bfhi(x1[k]), b0 = bflo(x2[k]), b1 = bfhi(x2[k]);
    float r10, r11, r20, r21;
    if (!bwd) {
      // out1 = x1*cos1 + (-x2)*sin1 ; out2 = x2*cos2 + x1*sin2
      r10 = bfround(a0 * bflo(c1[k])) + bfround(-b0 * bflo(s1[k]));
      r11 = bfround(a1 * bfhi(c1[k])) + bfround(-b1 * bfhi(s1[k]));
      r20 = bfround(b0 * bflo(c2[k])) + bfround(a0 * bflo(s2[k]));
      r21 = bfround(b1 * bfhi(c2[k])) + bfround(a1 * bfhi(s2[k]));
    } else {
      // dx1 = g1*cos1 + g2*sin2 ; dx2 = g2*cos2 - g1*sin1
      r10 = bfround(a0 * bflo(c1[k])) + bfround(b0 * bflo(s2[k]));
      r11 = bfround(a1 * bfhi(c1[k])) + bfround(b1 * bfhi(s2[k]));
      r20 = bfround(b0 * bflo(c2[k])) - bfround(a0 * bflo(s1[k]));
      r21 = bfround(b1 * bfhi(c2[k])) - bfround(a1 * bfhi(s1[k]));
    }
    o1[k] = pack2bf(r10, r11);
    o2[k] = pack2bf(r20, r21);
  }
  *(u32x4*)p = o1;
  *(u32x4*)(p + half) = o2;
}

// ------------------------------------------------------------------ SwiGLU
__device__ __forceinline__ float silu_f(float g) { return fast_silu(g); }

// seg_valid != NULL: rows are grouped in segments of seg_rows (MoE capacity slabs); rows at or past
// seg_valid[segment] are NOT read and are written as zeros (keeps dead capacity slots finite).
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                        bf16_t* __restrict__ out, long long rows, int I,
                                                        int ld_g, int ld_u, int ld_o, int seg_rows,
                                                        const int* __restrict__ seg_valid) {
  const int nch = I >> 3;
  const long long total = rows * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long r = id / nch; const int c = (int)(id - r * nch) * 8;
    if (seg_valid && (int)(r % seg_rows) >= seg_valid[r / seg_rows]) {
      *(u32x4*)(out + r * ld_o + c) = (u32x4){0u, 0u, 0u, 0u};
      continue;
    }
    const u32x4 g = *(const u32x4*)(gate + r * ld_g + c);
    const u32x4 u = *(const u32x4*)(up + r * ld_u + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack2bf(bfround(silu_f(bflo(g[k]))) * bflo(u[k]), bfround(silu_f(bfhi(g[k]))) * bfhi(u[k]));
    *(u32x4*)(out + r * ld_o + c) = o;
  }
}

// dgate = dact * up * silu'(gate) ; dup = dact * silu(gate).  dgate/dup may alias gate/up.
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* gate,
                                                        const bf16_t* up, bf16_t* dgate, bf16_t* dup,
                                                        long long rows, int I, int ld_d, int ld_g, int ld_u,
                                                        int ld_dg, int ld_du, int seg_rows,
                                                        const int* __restrict__ seg_valid) {
  const int nch = I >> 3;
  const long long total = rows * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long r = id / nch; const int c = (int)(id - r * nch) * 8;
    if (seg_valid && (int)(r % seg_rows) >= seg_valid[r / seg_rows]) {
      *(u32x4*)(dgate + r * ld_dg + c) = (u32x4){0u, 0u, 0u, 0u};
      *(u32x4*)(dup + r * ld_du + c) = (u32x4){0u, 0u, 0u, 0u};
      continue;
    }
    const u32x4 d = *(const u32x4*)(dact + r * ld_d + c);
    const u32x4 g = *(const u32x4*)(gate + r * ld_g + c);
    const u32x4 u = *(const u32x4*)(up + r * ld_u + c);
    u32x4 og, ou;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg[2] = {bflo(g[k]), bfhi(g[k])}, uu[2] = {bflo(u[k]), bfhi(u[k])}, dd[2] = {bflo(d[k]), bfhi(d[k])};
      float rg[2], ru[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sg = fast_sigmoid(gg[e]);
        const float si = gg[e] * sg;
        rg[e] = dd[e] * uu[e] * (sg * (1.f + gg[e] * (1.f - sg)));
        ru[e] = dd[e] * si;
      }
      og[k] = pack2bf(rg[0], rg[1]);
      ou[k] = pack2bf(ru[0], ru[1]);
    }
    *(u32x4*)(dgate + r * ld_dg + c) = og;
    *(u32x4*)(dup + r * ld_du + c) = ou;
  }
}

// ------------------------------------------------------------------ exact GELU (mm_projector)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long n8) {
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < n8; id += (long long)gridDim.x * 256) {
    const u32x4 v = *(const u32x4*)(x + id * 8);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = bflo(v[k]), b = bfhi(v[k]);
      o[k] = pack2bf(0.5f * a * (1.f + erff(a * 0.70710678118654752f)), 0.5f * b * (1.f + erff(b * 0.70710678118654752f)));
    }
    *(u32x4*)(y + id * 8) = o;
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                      bf16_t* __restrict__ dx, long long n8) {
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < n8; id += (long long)gridDim.x * 256) {
    const u32x4 v = *(const u32x4*)(x + id * 8);
    const u32x4 d = *(const u32x4*)(dy + id * 8);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float xx[2] = {bflo(v[k]), bfhi(v[k])}, dd[2] = {bflo(d[k]), bfhi(d[k])}, rr[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float cdf = 0.5f * (1.f + erff(xx[e] * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * xx[e] * xx[e]);
        rr[e] = dd[e] * (cdf + xx[e] * pdf);
      }
      o[k] = pack2bf(rr[0], rr[1]);
    }
    *(u32x4*)(dx + id * 8) = o;
  }
}

// ------------------------------------------------------------------ out = bf16(a + b)
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                 bf16_t* __restrict__ out, long long n8) {
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < n8; id += (long long)gridDim.x * 256) {
    const u32x4 x = *(const u32x4*)(a + id * 8), y = *(const u32x4*)(b + id * 8);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(bflo(x[k]) + bflo(y[k]), bfhi(x[k]) + bfhi(y[k]));
    *(u32x4*)(out + id * 8) = o;
  }
}

// ------------------------------------------------------------------ row gather from two tables
// idx >= 0: srcA[idx] ; idx <= -2: srcB[-(idx+2)] ; idx == -1: zeros.
// This is the embedding lookup + image-feature splice of llava_arch.py:236-318 done in one pass
// from a host-built index map, and (with the inverse map) its backward into the projector output.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ srcA, const bf16_t* __restrict__ srcB,
                                                         const int* __restrict__ idx, bf16_t* __restrict__ out,
                                                         long long rows, int H) {
  const int nch = H >> 3;
  const long long total = rows * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long r = id / nch; const int c = (int)(id - r * nch) * 8;
    const int ix = idx[r];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ix >= 0) v = *(const u32x4*)(srcA + (long long)ix * H + c);
    else if (ix <= -2) v = *(const u32x4*)(srcB + (long long)(-(ix + 2)) * H + c);
    *(u32x4*)(out + r * H + c) = v;
  }
}

// ------------------------------------------------------------------ ViT front end
// im2col for Conv2d(3, D, kernel=P, stride=P, bias=False): pixels [B,3,S,S] -> [B*(S/P)^2, Kpad],
// column = c*P*P + py*P + px (the conv weight's flatten order), zero-padded to Kpad.
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ pix, bf16_t* __restrict__ out,
                                                    int B, int S, int P, int Kpad) {
  const int G = S / P;
  const long long total = (long long)B * G * G * Kpad;
  const int K = 3 * P * P;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int k = (int)(id % Kpad);
    const long long pr = id / Kpad;
    bf16_t v = 0;
    if (k < K) {
      const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
      const int gx = (int)(pr % G), gy = (int)((pr / G) % G);
      const long long b = pr / ((long long)G * G);
      v = pix[((b * 3 + c) * S + (gy * P + py)) * (long long)S + gx * P + px];
    }
    out[id] = v;
  }
}

// tokens[b,0,:] = cls + pos[0];  tokens[b,1+p,:] = patch[b,p,:] + pos[1+p]
__global__ __launch_bounds__(256) void vit_embed_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                                       const bf16_t* __restrict__ pos, bf16_t* __restrict__ out,
                                                       int B, int NP, int D) {
  const int nch = D >> 3;
  const long long total = (long long)B * (NP + 1) * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int c = (int)(id % nch) * 8;
    const long long tr = id / nch;
    const int tok = (int)(tr % (NP + 1));
    const long long b = tr / (NP + 1);
    const u32x4 pv = *(const u32x4*)(pos + (long long)tok * D + c);
    u32x4 xv;
    if (tok == 0) xv = *(const u32x4*)(cls + c);
    else xv = *(const u32x4*)(patch + (b * NP + tok - 1) * (long long)D + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(bflo(xv[k]) + bflo(pv[k]), bfhi(xv[k]) + bfhi(pv[k]));
    *(u32x4*)(out + tr * D + c) = o;
  }
}

// ------------------------------------------------------------------ fused AdamW (decoupled decay)
// fp32 master/m/v, fp32 grad, bf16 working copy refreshed in the same pass.
// ---------------------------------------------------------------- weight gradients of the "row" parameters
// RMSNorm scale (Qwen2RMSNorm, qwen2/modeling_qwen2.py:92-97: y = w * bf16(h * rstd)):  dw[c] = sum_t dy[t,c] * bf16(h[t,c] rstd[t]).
// Deterministic two-stage column reduction: NSPLIT row slices -> partials[NSPLIT][H] -> fixed-order sum (+= dw).
// Only runs when a norm scale is trainable (the distillation shells freeze them).
#define DW_NSPLIT 64
__global__ __launch_bounds__(256) void rmsnorm_dw_partial_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h,
                                                                const float* __restrict__ rstd, float* __restrict__ part,
                                                                int T, int H) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;           // a bf16 pair per thread: 1 KiB contiguous per row and block
  if (c >= H) return;
  float a0 = 0.f, a1 = 0.f;
  for (int t = blockIdx.y; t < T; t += DW_NSPLIT) {
    const uint32_t d = *(const uint32_t*)(dy + (long long)t * H + c), x = *(const uint32_t*)(h + (long long)t * H + c);
    const float r = rstd[t];
    a0 += bflo(d) * bfround(bflo(x) * r);
    a1 += bfhi(d) * bfround(bfhi(x) * r);
  }
  part[(long long)blockIdx.y * H + c] = a0;
  part[(long long)blockIdx.y * H + c + 1] = a1;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nsplit, int H,
                                                          int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= H) return;
  float a = 0.f;
  for (int s = 0; s < nsplit; ++s) a += part[(long long)s * H + c];
  out[c] = accumulate ? out[c] + a : a;
}
// Embedding table (embed_tokens): dW[idx[r], :] += d_embeds[r, :] for idx[r] >= 0 (text rows of the spliced sequence).
// fp32 atomics: repeated token ids collide, so the summation ORDER is not fixed run to run (values agree to fp32 rounding).
__global__ __launch_bounds__(256) void embed_wgrad_kernel(const bf16_t* __restrict__ d, const int* __restrict__ idx,
                                                         float* __restrict__ dW, long long rows, int H) {
  const int nch = H >> 1;
  const long long total = rows * nch;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const long long r = id / nch; const int c = (int)(id - r * nch) * 2;
    const int ix = idx[r];
    if (ix < 0) continue;
    const uint32_t v = *(const uint32_t*)(d + r * H + c);
    unsafeAtomicAdd(dW + (long long)ix * H + c, bflo(v));
    unsafeAtomicAdd(dW + (long long)ix * H + c + 1, bfhi(v));
  }
}

// Matches torch.optim.AdamW (HF `adamw_torch`, reference config/args.py:78) step arithmetic.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, bf16_t* __restrict__ param,
                                                   float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2, float gscale, int zero_grad,
                                                   const float* __restrict__ dev_scale) {
  if (dev_scale) gscale *= dev_scale[0];     // gradient-clipping coefficient computed on the device (no host sync)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = grad[i] * gscale;
    if (zero_grad) grad[i] = 0.f;            // the gradient buffer is consumed: no separate memset before the next step
    float p = master[i];
    p *= (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p -= (lr / bc1) * mi / denom;
    master[i] = p;
    param[i] = f2bf(p);
  }
}

// Global gradient norm for clipping (HF Trainer max_grad_norm / DeepSpeed gradient_clipping, reference
// config/dpconfig/zero2*.json "gradient_clipping": "auto"): deterministic two-stage sum of squares.  Stage 1: SUMSQ_BLOCKS
// blocks grid-stride over the span, one partial each; stage 2: one block adds the partials in a fixed order into out[0]
// (accumulating, so several spans chain into one scalar).
#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  // up to 3 leading elements bring the pointer to a 16-byte boundary (block 0), then 16-byte loads, then the tail
  const int head = (int)min((long long)(((16 - ((uintptr_t)x & 15)) & 15) >> 2), n);
  if (blockIdx.x == 0 && threadIdx.x < head) { const float t = x[threadIdx.x]; a += t * t; }
  x += head; n -= head;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x - 64 < (n & 3)) { const float t = x[(n4 << 2) + threadIdx.x - 64]; a += t * t; }
  a = block_sum<4>(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
  a = block_sum<4>(a, red);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + a : a;
}
// coef = min(1, max_norm / (sqrt(sumsq) * norm_scale + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float norm_scale, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float nrm = sqrtf(sumsq[0]) * norm_scale;
  coef[0] = fminf(1.f, max_norm / (nrm + 1e-6f));
  if (norm_out) norm_out[0] = nrm;
}
// fp32 -> bf16 (round to nearest even) and back, for exchanging gradients in bf16 like the reference's bf16 engine
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    *(u32x2*)(y + 4 * i) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = f2bf(x[(n4 << 2) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const u32x2 v = *(const u32x2*)(x + 4 * i);
    *(f32x4*)(y + 4 * i) = (f32x4){bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = bf2f(x[(n4 << 2) + threadIdx.x]);
}

static inline int grid_for(long long work, int cap = 256 * 16) {
  long long b = (work + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

extern "C" {

int lmod_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd,
                     int T, int H, float eps, hipStream_t stream) {
  if (!x || !w || !y || T < 0 || H <= 0 || (H & 7) || H > 64 * MAXCH * 8) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x,
                     (const bf16_t*)res, (const bf16_t*)w, (bf16_t*)h_out, (bf16_t*)y, rstd, T, H, eps);
  return lmod_launch_status();
}

int lmod_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres,
                     void* dh, int T, int H, hipStream_t stream) {
  if (!dy || !h || !w || !rstd || !dh || T < 0 || H <= 0 || (H & 7) || H > 64 * MAXCH * 8) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)h, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dh, T, H);
  return lmod_launch_status();
}

int lmod_rmsnorm_dw(const void* dy, const void* h, const float* rstd, float* dw, float* workspace, int T, int H,
                    int accumulate, hipStream_t stream) {
  if (!dy || !h || !rstd || !dw || !workspace || T < 0 || H <= 0 || (H & 7)) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(rmsnorm_dw_partial_kernel, dim3((H / 2 + 255) / 256, DW_NSPLIT), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)h, rstd, workspace, T, H);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((H + 255) / 256), dim3(256), 0, stream, workspace, dw, min(DW_NSPLIT, T), H, accumulate);
  return lmod_launch_status();
}

int lmod_embed_wgrad(const void* d_embeds, const int* idx, float* dW, long long rows, int H, hipStream_t stream) {
  if (!d_embeds || !idx || !dW || rows < 0 || H <= 0 || (H & 7)) return LMOD_EINVAL;
  if (rows == 0) return LMOD_OK;
  hipLaunchKernelGGL(embed_wgrad_kernel, dim3(grid_for(rows * (H >> 1))), dim3(256), 0, stream, (const bf16_t*)d_embeds, idx, dW,
                     rows, H);
  return lmod_launch_status();
}

int lmod_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int T, int H, float eps,
                       hipStream_t stream) {
  if (!x || !w || !b || !y || T < 0 || H <= 0 || (H & 7) || H > 64 * MAXCH * 8) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x,
                     (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, T, H, eps);
  return lmod_launch_status();
}

int lmod_rope(void* buf, const void* cos_t, const void* sin_t, const int* pos, int T, int nheads, int hd,
              int ld, int backward, hipStream_t stream) {
  if (!buf || !cos_t || !sin_t || !pos || T < 0 || nheads <= 0 || hd <= 0 || (hd & 15) || (ld & 7) ||
      ld < nheads * hd) return LMOD_EINVAL;
  if (T == 0) return LMOD_OK;
  const long long total = (long long)T * nheads * (hd >> 4);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (bf16_t*)buf,
                     (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, T, nheads, hd, ld, backward);
  return lmod_launch_status();
}

int lmod_swiglu_fwd(const void* gate, const void* up, void* out, long long rows, int I, int ld_gate, int ld_up,
                    int ld_out, int seg_rows, const int* seg_valid, hipStream_t stream) {
  if (!gate || !up || !out || rows < 0 || I <= 0 || (I & 7) || (ld_gate & 7) || (ld_up & 7) || (ld_out & 7) ||
      (seg_valid && seg_rows <= 0)) return LMOD_EINVAL;
  if (rows == 0) return LMOD_OK;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (I >> 3))), dim3(256), 0, stream, (const bf16_t*)gate,
                     (const bf16_t*)up, (bf16_t*)out, rows, I, ld_gate, ld_up, ld_out, seg_rows, seg_valid);
  return lmod_launch_status();
}

int lmod_swiglu_bwd(const void* dact, const void* gate, const void* up, void* dgate, void* dup, long long rows,
                    int I, int ld_dact, int ld_gate, int ld_up, int ld_dgate, int ld_dup, int seg_rows,
                    const int* seg_valid, hipStream_t stream) {
  if (!dact || !gate || !up || !dgate || !dup || rows < 0 || I <= 0 || (I & 7) || (ld_dact & 7) ||
      (ld_gate & 7) || (ld_up & 7) || (ld_dgate & 7) || (ld_dup & 7) || (seg_valid && seg_rows <= 0)) return LMOD_EINVAL;
  if (rows == 0) return LMOD_OK;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (I >> 3))), dim3(256), 0, stream, (const bf16_t*)dact,
                     (const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)dgate, (bf16_t*)dup, rows, I, ld_dact,
                     ld_gate, ld_up, ld_dgate, ld_dup, seg_rows, seg_valid);
  return lmod_launch_status();
}

int lmod_gelu_fwd(const void* x, void* y, long long n, hipStream_t stream) {
  if (!x || !y || n < 0 || (n & 7)) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n >> 3);
  return lmod_launch_status();
}

int lmod_gelu_bwd(const void* dy, const void* x, void* dx, long long n, hipStream_t stream) {
  if (!dy || !x || !dx || n < 0 || (n & 7)) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)x, (bf16_t*)dx, n >> 3);
  return lmod_launch_status();
}

int lmod_add_bf16(const void* a, const void* b, void* out, long long n, hipStream_t stream) {
  if (!a || !b || !out || n < 0 || (n & 7)) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)out, n >> 3);
  return lmod_launch_status();
}

int lmod_gather_rows(const void* srcA, const void* srcB, const int* idx, void* out, long long rows, int H,
                     hipStream_t stream) {
  if (rows < 0 || H <= 0 || (H & 7)) return LMOD_EINVAL;
  if (rows == 0) return LMOD_OK;                  // an empty gather has no pointers to check (an expert-parallel rank with no live rows)
  if (!idx || !out) return LMOD_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (H >> 3))), dim3(256), 0, stream, (const bf16_t*)srcA,
                     (const bf16_t*)srcB, idx, (bf16_t*)out, rows, H);
  return lmod_launch_status();
}

int lmod_im2col_patch(const void* pixels, void* out, int B, int image_size, int patch, int Kpad, hipStream_t stream) {
  if (!pixels || !out || B < 0 || patch <= 0 || image_size % patch || Kpad < 3 * patch * patch || (Kpad & 7))
    return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  const int G = image_size / patch;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((long long)B * G * G * Kpad)), dim3(256), 0, stream,
                     (const bf16_t*)pixels, (bf16_t*)out, B, image_size, patch, Kpad);
  return lmod_launch_status();
}

int lmod_vit_embed(const void* patch_emb, const void* cls, const void* pos, void* out, int B, int n_patches, int D,
                   hipStream_t stream) {
  if (!patch_emb || !cls || !pos || !out || B < 0 || n_patches <= 0 || D <= 0 || (D & 7)) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  hipLaunchKernelGGL(vit_embed_kernel, dim3(grid_for((long long)B * (n_patches + 1) * (D >> 3))), dim3(256), 0,
                     stream, (const bf16_t*)patch_emb, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, B,
                     n_patches, D);
  return lmod_launch_status();
}

int lmod_adamw_step(float* master, void* param_bf16, float* grad, float* m, float* v, long long n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad,
                    const float* dev_scale, hipStream_t stream) {
  if (!master || !param_bf16 || !grad || !m || !v || n < 0 || step < 1) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, master, (bf16_t*)param_bf16, grad, m, v,
                     n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, zero_grad, dev_scale);
  return lmod_launch_status();
}

int lmod_sumsq_f32(const float* x, long long n, float* partials, float* out, int accumulate, hipStream_t stream) {
  if (!partials || !out || n < 0 || (n > 0 && !x)) return LMOD_EINVAL;
  if (((uintptr_t)x & 3)) return LMOD_EINVAL;
  const int nb = n == 0 ? 0 : grid_for((n + 3) / 4, SUMSQ_BLOCKS);
  if (nb) hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, x, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, nb, out, accumulate);
  return lmod_launch_status();
}

int lmod_clip_coef(const float* sumsq, float norm_scale, float max_norm, float* coef, float* norm_out, hipStream_t stream) {
  if (!sumsq || !coef || !(max_norm > 0.f)) return LMOD_EINVAL;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, norm_scale, max_norm, coef, norm_out);
  return lmod_launch_status();
}

int lmod_cast_f32_bf16(const void* src, void* dst, long long n, int to_bf16, hipStream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return LMOD_EINVAL;
  if (to_bf16) hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n);
  return lmod_launch_status();
}

}  // extern "C"
