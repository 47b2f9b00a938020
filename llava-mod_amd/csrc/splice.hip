// Device-side index build for the multimodal splice and for the loss-row plan.
//
// prepare_inputs_labels_for_multimodal (reference llava_arch.py:155-334) walks the batch in python: strip pads by mask,
// split every sample at its <image> (-200) tokens (with a `.tolist()` host sync, :247), concatenate text embeddings and
// image features, pad to the batch maximum, rebuild labels / mask.  Here the same layout is two launches over int
// tensors that are already on the device: per-sample prefix sums give every token its spliced position.
//   lmod_splice_count : new length and number of image slots consumed per sample      (the host reads B ints: S' = max)
//   lmod_splice_fill  : idx[B*S'] (>=0 embedding row | <=-2 projector row -(i+2) | -1 pad), labels', mask', inverse map
// and, for the loss head, which rows of the [B*S'] hidden states carry loss (AlignTrainer.compute_align_loss's unshifted
// mask / the shifted CE rows, align_trainer.py:503-528, llava_qwen2_moe.py:413-421):
//   lmod_lossplan_count / lmod_lossplan_fill : compaction of those rows in sample-major order.
// One workgroup per sample; tokens are scanned in chunks of 256 with wave ballots.
#include "common.h"

#define IMG_TOKEN (-200)
#define IGNORE (-100)

// exclusive prefix of two predicates over the 256 threads of the block; returns this thread's (ea, eb) and the block totals
__device__ __forceinline__ void scan2(bool a, bool b, int& ea, int& eb, int& ta, int& tb, int (*sh)[4]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long ma = __ballot(a), mb = __ballot(b), below = (1ull << lane) - 1ull;
  __syncthreads();
  if (lane == 0) { sh[0][w] = __popcll(ma); sh[1][w] = __popcll(mb); }
  __syncthreads();
  ea = __popcll(ma & below); eb = __popcll(mb & below); ta = 0; tb = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    if (x < w) { ea += sh[0][x]; eb += sh[1][x]; }
    ta += sh[0][x]; tb += sh[1][x];
  }
}

__global__ __launch_bounds__(256) void splice_count_kernel(const long long* __restrict__ ids, const unsigned char* __restrict__ mask,
                                                          int T, int P, int max_len, int* __restrict__ lens,
                                                          int* __restrict__ nimg) {
  __shared__ int sh[2][4];
  const int b = blockIdx.x;
  int nv = 0, ni = 0;
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool v = t < T && (!mask || mask[(long long)b * T + t]);
    const bool im = v && ids[(long long)b * T + t] == IMG_TOKEN;
    int ea, eb, ta, tb;
    scan2(v, im, ea, eb, ta, tb, sh);
    nv += ta; ni += tb;
  }
  if (threadIdx.x == 0) {
    int L = nv - ni + ni * P;
    if (max_len > 0) L = min(L, max_len);
    lens[b] = L;
    nimg[b] = max(ni, 1);                    // a sample without <image> still consumes one image slot (:238-246)
  }
}

__global__ __launch_bounds__(256) void splice_fill_kernel(const long long* __restrict__ ids, const unsigned char* __restrict__ mask,
                                                         const long long* __restrict__ labels, int T, int P, int S,
                                                         const int* __restrict__ lens, const int* __restrict__ nimg,
                                                         int* __restrict__ idx, long long* __restrict__ new_labels,
                                                         unsigned char* __restrict__ new_mask, int* __restrict__ inv) {
  __shared__ int sh[2][4];
  const int b = blockIdx.x;
  const int L = lens[b];
  int k0 = 0;                                // first image slot of this sample
  for (int x = 0; x < b; ++x) k0 += nimg[x];
  int nv = 0, ni = 0;
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool v = t < T && (!mask || mask[(long long)b * T + t]);
    const long long id = v ? ids[(long long)b * T + t] : 0;
    const bool im = v && id == IMG_TOKEN;
    int ea, eb, ta, tb;
    scan2(v, im, ea, eb, ta, tb, sh);
    if (v) {
      const int nvb = nv + ea, nib = ni + eb;                 // valid tokens / image tokens before this one
      const int p = (nvb - nib) + nib * P;                    // spliced position
      if (!im) {
        if (p < L) {
          idx[(long long)b * S + p] = (int)id;
          new_labels[(long long)b * S + p] = labels ? labels[(long long)b * T + t] : IGNORE;
          if (new_mask) new_mask[(long long)b * S + p] = 1;
        }
      } else {
        const int k = k0 + nib;
        for (int j = 0; j < P; ++j) {
          if (p + j < L) {
            idx[(long long)b * S + p + j] = -(k * P + j + 2);
            new_labels[(long long)b * S + p + j] = IGNORE;
            if (new_mask) new_mask[(long long)b * S + p + j] = 1;
            inv[k * P + j] = b * S + p + j;
          }
        }
      }
    }
    nv += ta; ni += tb;
  }
  for (int p = L + threadIdx.x; p < S; p += 256) {            // right padding
    idx[(long long)b * S + p] = -1;
    new_labels[(long long)b * S + p] = IGNORE;
    if (new_mask) new_mask[(long long)b * S + p] = 0;
  }
}

// need[t] = kd[t] | ce[t];  kd[t] = all_tokens ? 1 : labels[t] != -100;  ce[t] = t + 1 < S && labels[t + 1] != -100
__global__ __launch_bounds__(256) void lossplan_count_kernel(const long long* __restrict__ labels, int S, int kd_rows, int ce_rows,
                                                            int all_tokens, int* __restrict__ counts) {
  __shared__ int sh[2][4];
  const int b = blockIdx.x;
  int n = 0;
  for (int t0 = 0; t0 < S; t0 += 256) {
    const int t = t0 + threadIdx.x;
    bool need = false;
    if (t < S) {
      const bool kd = kd_rows && (all_tokens || labels[(long long)b * S + t] != IGNORE);
      const bool ce = ce_rows && t + 1 < S && labels[(long long)b * S + t + 1] != IGNORE;
      need = kd || ce;
    }
    int ea, eb, ta, tb;
    scan2(need, false, ea, eb, ta, tb, sh);
    n += ta;
  }
  if (threadIdx.x == 0) counts[b] = n;
}

__global__ __launch_bounds__(256) void lossplan_fill_kernel(const long long* __restrict__ labels, int S, int B, int kd_rows, int ce_rows,
                                                           int all_tokens, const int* __restrict__ counts, int* __restrict__ row_idx,
                                                           int* __restrict__ inv_row, float* __restrict__ kd_w,
                                                           float* __restrict__ ce_w, int* __restrict__ ce_label,
                                                           int* __restrict__ seg_off, int* __restrict__ seg_id) {
  __shared__ int sh[2][4];
  const int b = blockIdx.x;
  int base = 0;
  for (int x = 0; x < b; ++x) base += counts[x];
  if (threadIdx.x == 0) {
    seg_off[b] = base;
    if (b == B - 1) seg_off[B] = base + counts[b];
  }
  int n = 0;
  for (int t0 = 0; t0 < S; t0 += 256) {
    const int t = t0 + threadIdx.x;
    bool kd = false, ce = false;
    long long nxt = IGNORE;
    if (t < S) {
      kd = kd_rows && (all_tokens || labels[(long long)b * S + t] != IGNORE);
      if (t + 1 < S) nxt = labels[(long long)b * S + t + 1];
      ce = ce_rows && nxt != IGNORE;
    }
    const bool need = kd || ce;
    int ea, eb, ta, tb;
    scan2(need, false, ea, eb, ta, tb, sh);
    if (t < S) {
      if (need) {
        const int r = base + n + ea;
        row_idx[r] = b * S + t;
        inv_row[(long long)b * S + t] = r;
        kd_w[r] = kd ? 1.f : 0.f;
        ce_w[r] = ce ? 1.f : 0.f;
        ce_label[r] = (nxt != IGNORE) ? (int)nxt : -1;      // the shifted label itself, whether or not CE rows are selected
        seg_id[r] = b;
      } else {
        inv_row[(long long)b * S + t] = -1;
      }
    }
    n += ta;
  }
}

extern "C" {

int lmod_splice_count(const long long* input_ids, const unsigned char* attention_mask, int B, int T, int n_patches,
                      int max_length, int* lens, int* n_images, hipStream_t stream) {
  if (B < 0 || T <= 0 || n_patches <= 0) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  if (!input_ids || !lens || !n_images) return LMOD_EINVAL;
  hipLaunchKernelGGL(splice_count_kernel, dim3(B), dim3(256), 0, stream, input_ids, attention_mask, T, n_patches, max_length,
                     lens, n_images);
  return lmod_launch_status();
}

int lmod_splice_fill(const long long* input_ids, const unsigned char* attention_mask, const long long* labels, int B, int T,
                     int n_patches, int S, const int* lens, const int* n_images, int* idx, long long* new_labels,
                     unsigned char* new_mask, int* inv_idx, hipStream_t stream) {
  if (B < 0 || T <= 0 || n_patches <= 0 || S <= 0) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  if (!input_ids || !lens || !n_images || !idx || !new_labels || !inv_idx) return LMOD_EINVAL;
  hipLaunchKernelGGL(splice_fill_kernel, dim3(B), dim3(256), 0, stream, input_ids, attention_mask, labels, T, n_patches, S, lens,
                     n_images, idx, new_labels, new_mask, inv_idx);
  return lmod_launch_status();
}

int lmod_lossplan_count(const long long* labels, int B, int S, int kd_rows, int ce_rows, int all_tokens, int* counts,
                        hipStream_t stream) {
  if (B < 0 || S <= 0) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  if (!labels || !counts) return LMOD_EINVAL;
  hipLaunchKernelGGL(lossplan_count_kernel, dim3(B), dim3(256), 0, stream, labels, S, kd_rows, ce_rows, all_tokens, counts);
  return lmod_launch_status();
}

int lmod_lossplan_fill(const long long* labels, int B, int S, int kd_rows, int ce_rows, int all_tokens, const int* counts,
                       int* row_idx, int* inv_row_idx, float* kd_w, float* ce_w, int* ce_label, int* seg_off, int* seg_id,
                       hipStream_t stream) {
  if (B < 0 || S <= 0) return LMOD_EINVAL;
  if (B == 0) return LMOD_OK;
  if (!labels || !counts || !row_idx || !inv_row_idx || !kd_w || !ce_w || !ce_label || !seg_off || !seg_id) return LMOD_EINVAL;
  hipLaunchKernelGGL(lossplan_fill_kernel, dim3(B), dim3(256), 0, stream, labels, S, B, kd_rows, ce_rows, all_tokens, counts,
                     row_idx, inv_row_idx, kd_w, ce_w, ce_label, seg_off, seg_id);
  return lmod_launch_status();
}

}  // extern "C"
