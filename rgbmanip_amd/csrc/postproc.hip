// AdaPose post-processing on the GPU (gfx950): replaces the numpy code the reference runs on the
// host after every network call
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:76-119  (compute_scale[_and_translation])
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:40-74   (get_3d_bbox, transform_coordinates_3d)
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:354-374 (bbox -> world, default_bbox)
// One workgroup (1024 threads) per pose.  The O(P^2) pair ratios are never stored: the exact median is
// found by a 12-bit-digit radix select over the order-preserving bit pattern of the positive fp64
// ratios, recomputing the ratios each pass from LDS-resident points (only i<j pairs: the reference's
// flattened P x P list holds every ratio twice and i==j is excluded by the nocs>0.01 filter, so both
// lists have the same median).  Arithmetic follows the reference's dtypes: NOCS distances in fp32,
// camera-space distances / ratios / median in fp64 (contraction disabled where numpy rounds twice).
#include "common.h"
#include "kernels.h"
#include "bbox_emit.h"

// numpy rounds every elementwise op separately: forbid a*b+c -> fma fusion in this file
#pragma clang fp contract(off)

namespace rgbm {

constexpr int PP_MAXP = 1024;
constexpr int PP_THREADS = 1024;
constexpr int PP_CAND = 3072;

__device__ __forceinline__ bool pair_ratio(const double* cx, const double* cy, const double* cz, const float* nx,
                                           const float* ny, const float* nz, int i, int j, double& ratio) {
  // plain IEEE operators (this file is built with -ffp-contract=off; sqrtf/sqrt/÷ are correctly rounded,
  // whereas HIP's __fsqrt_rn/__fdiv_rn wrappers lower to the approximate native instructions)
  const float dx = nx[i] - nx[j], dy = ny[i] - ny[j], dz = nz[i] - nz[j];
  const float nd = sqrtf((dx * dx + dy * dy) + dz * dz);
  if (!(nd > 0.01f)) return false;
  const double ex = cx[i] - cx[j], ey = cy[i] - cy[j], ez = cz[i] - cz[j];
  const double rd = sqrt((ex * ex + ey * ey) + ez * ez);
  if (!(rd < 0.3)) return false;
  ratio = rd / (double)nd;
  return true;
}

// map (row-pair r, q) -> (i, j), i<j, covering all P(P-1)/2 pairs with P-1 pairs per row pair
__device__ __forceinline__ void pair_ij(int P, int r, int q, int& i, int& j) {
  if (q < P - 1 - r) { i = r; j = r + 1 + q; }
  else { i = P - 1 - r; j = q + 1; }
}

__device__ double block_sum(double v, double* red) {
  // 1024 threads -> deterministic tree in LDS
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = PP_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// Split form for small batches (one workgroup per pose leaves 255 CUs idle for 1.1 ms at B = 1: 37 % of a B = 1 forward, 28 % at
// B = 8).  The three passes over the 523 776 pairs that the selection normally needs — the two 12-bit histograms and the collection
// of the selected bucket's keys — are cut into G slices per pose, one workgroup each (pp_split_kernel, STAGE 0 / 1 / 2; the slices
// of a pose add their histograms / keys into global scratch), and the finishing kernel (postprocess_kernel<true>, one workgroup per
// pose) picks the state up from there: same integer histograms, same bucket, same key set, hence the same median bit for bit.  If a
// bucket holds more than PP_CAND keys the finishing kernel simply continues with the remaining radix passes on its own.
struct PPScratch {                    // per pose, in caller-provided device memory (zeroed in front of every call)
  unsigned hist0[4096], hist1[4096];
  unsigned long long cand[PP_CAND];
  unsigned ncand, pad[3];
};

template <bool PRE>
__global__ __launch_bounds__(PP_THREADS) void postprocess_kernel(
    const float* __restrict__ nocs /*[B,P,3]*/, const float* __restrict__ depth /*[B,P]*/, const float* __restrict__ rot /*[B,9]*/,
    const int* __restrict__ choose /*[B,P]*/, const double* __restrict__ Kc /*[B,9]*/, const double* __restrict__ E1 /*[B,16]*/,
    double* __restrict__ bbox /*[B,8,3]*/, double* __restrict__ ts_out /*[B,4]: t(3), s*/, int* __restrict__ valid, int P, int img,
    const PPScratch* __restrict__ pre) {
  __shared__ double cx[PP_MAXP], cy[PP_MAXP], cz[PP_MAXP];
  __shared__ float nx[PP_MAXP], ny[PP_MAXP], nz[PP_MAXP];
  __shared__ double red[PP_THREADS];
  __shared__ unsigned hist[4096];
  __shared__ unsigned long long sel_prefix;
  __shared__ unsigned sel_rank, sel_lt, sel_eq, total_cnt;
  __shared__ unsigned wsum[PP_THREADS / 64];
  __shared__ unsigned long long cand[PP_CAND];   // keys of the selected 24-bit bucket (short cut after two radix passes)
  __shared__ unsigned ncand, cand_lt;
  __shared__ unsigned long long cand_med, cand_below;
  __shared__ float hmax[3];

  const int b = blockIdx.x, t = threadIdx.x;
  const double fx = Kc[b * 9 + 0], fy = Kc[b * 9 + 4], pcx = Kc[b * 9 + 2], pcy = Kc[b * 9 + 5];
  if (t < P) {
    const int ch = choose[(long long)b * P + t];
    const double u = (double)(ch % img), v = (double)(ch / img);
    const double z = (double)depth[(long long)b * P + t];
    cx[t] = ((u - pcx) * z) / fx;
    cy[t] = ((v - pcy) * z) / fy;
    cz[t] = z;
    nx[t] = nocs[((long long)b * P + t) * 3 + 0];
    ny[t] = nocs[((long long)b * P + t) * 3 + 1];
    nz[t] = nocs[((long long)b * P + t) * 3 + 2];
  }
  if (t < 3) hmax[t] = 0.f;
  __syncthreads();

  // ---- exact median of the valid pair ratios: radix select, digits 12,12,12,12,12,4 bits --------
  const int npair_rows = P / 2;                  // row pairs (P even)
  const int per_row = P - 1;
  const int nthr_per_row = PP_THREADS / npair_rows;   // >=2 for P<=1024 ... 1024/512 = 2
  const int r = t % npair_rows, part = t / npair_rows;
  const int q_lo = (int)(((long long)per_row * part) / nthr_per_row);
  const int q_hi = (int)(((long long)per_row * (part + 1)) / nthr_per_row);
  const bool worker = part < nthr_per_row;

  unsigned long long prefix = 0ull;     // selected high bits so far (right-aligned)
  unsigned rank = 0;                    // rank to find within the current prefix bucket
  double med_hi = __builtin_nan(""), med_lo = __builtin_nan("");
  bool have = false, need_lower = false;
  const int shifts[6] = {52, 40, 28, 16, 4, 0};
  const int widths[6] = {12, 12, 12, 12, 12, 4};
  unsigned lt_total = 0;                // number of elements strictly below the selected bucket (overall)
  bool used_cand = false;
  for (int pass = 0; pass < 6; ++pass) {
    const int sh = shifts[pass], wd = widths[pass];
    if (PRE && pass < 2) {                 // the slices of pp_split_kernel have already counted this digit
      const unsigned* gh = pass == 0 ? pre[b].hist0 : pre[b].hist1;
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = gh[i];
    } else {
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
      __syncthreads();
      if (worker) {
        for (int q = q_lo; q < q_hi; ++q) {
          int i, j;
          pair_ij(P, r, q, i, j);
          double ratio;
          if (!pair_ratio(cx, cy, cz, nx, ny, nz, i, j, ratio)) continue;
          const unsigned long long key = (unsigned long long)__double_as_longlong(ratio);
          if (pass > 0 && (key >> (sh + wd)) != prefix) continue;
          atomicAdd(&hist[(unsigned)((key >> sh) & ((1u << wd) - 1u))], 1u);
        }
      }
    }
    __syncthreads();
    // bucket selection by all 1024 threads (4 bins each): block-wide exclusive scan of the bin counts, then the one
    // thread whose bins contain the wanted rank publishes the digit.  (A single thread walking 4096 bins per pass cost
    // ~0.9 ms of the 3.3 ms this kernel took for 256 poses.)
    {
      const int nb = 1 << wd;
      unsigned c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = (4 * t + k < nb) ? hist[4 * t + k] : 0u;
      const unsigned tsum = c[0] + c[1] + c[2] + c[3];
      unsigned incl = tsum;
      const int ln = t & 63, wv = t >> 6;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (ln >= off) incl += o;
      }
      if (ln == 63) wsum[wv] = incl;
      __syncthreads();
      unsigned wbase = 0, total = 0;
      for (int k = 0; k < PP_THREADS / 64; ++k) { const unsigned w = wsum[k]; if (k < wv) wbase += w; total += w; }
      const unsigned excl = wbase + incl - tsum;
      const unsigned rk = pass == 0 ? total / 2 : sel_rank;      // pass 0: upper middle (0-based) of the i<j list
      __syncthreads();                                            // everybody has read sel_rank / wsum
      if (pass == 0 && t == 0) total_cnt = total;
      // rank inside my bins, or (defensive, cannot happen for a consistent histogram) past the end: last bin
      const bool mine = (rk >= excl && rk < excl + tsum) || (rk >= total && 4 * t <= nb - 1 && nb - 1 < 4 * t + 4);
      if (mine) {
        unsigned acc = excl;
        int dsel = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (dsel < 0) {
            if (acc + c[k] > rk) dsel = 4 * t + k; else acc += c[k];
          }
        }
        if (dsel < 0) { dsel = nb - 1; acc = total - hist[nb - 1]; }
        sel_prefix = (pass == 0 ? 0ull : (sel_prefix << wd)) | (unsigned long long)dsel;
        if (pass == 0) sel_lt = acc; else sel_lt += acc;
        sel_rank = rk - acc;
        sel_eq = hist[dsel];
      }
    }
    __syncthreads();
    prefix = sel_prefix;
    rank = sel_rank;
    lt_total = sel_lt;
    if (total_cnt == 0) break;
    // Short cut: after two passes the bucket is 2^-12 wide in relative terms and holds a few hundred of the 523 776 ratios.
    // Collect its keys once and finish the selection on that list instead of four more passes over all pairs.
    if (pass == 1 && sel_eq <= (unsigned)PP_CAND) {
      if (t == 0) { ncand = PRE ? sel_eq : 0u; cand_lt = 0u; cand_below = 0ull; }
      __syncthreads();
      if (PRE) {                                   // the bucket's keys, collected by the slices (in any order: only values matter below)
        for (unsigned c = t; c < sel_eq; c += PP_THREADS) cand[c] = pre[b].cand[c];
      } else if (worker) {
        for (int q = q_lo; q < q_hi; ++q) {
          int i, j;
          pair_ij(P, r, q, i, j);
          double ratio;
          if (!pair_ratio(cx, cy, cz, nx, ny, nz, i, j, ratio)) continue;
          const unsigned long long key = (unsigned long long)__double_as_longlong(ratio);
          if ((key >> 40) == prefix) cand[atomicAdd(&ncand, 1u)] = key;
        }
      }
      __syncthreads();
      const unsigned m = ncand;                    // == sel_eq
      for (unsigned c = t; c < m; c += PP_THREADS) {
        const unsigned long long kc = cand[c];
        unsigned lt = 0, eq_before = 0;
        for (unsigned o = 0; o < m; ++o) { const unsigned long long ko = cand[o]; lt += ko < kc; eq_before += (ko == kc) & (o < c); }
        if (lt + eq_before == rank) { cand_med = kc; cand_lt = lt; }     // exactly one candidate has this rank
      }
      __syncthreads();
      const unsigned long long kmed = cand_med;
      unsigned long long below = 0ull;             // largest key strictly below the median inside the bucket (0 = none)
      for (unsigned c = t; c < m; c += PP_THREADS) { const unsigned long long kc = cand[c]; if (kc < kmed && kc > below) below = kc; }
      if (below) atomicMax(&cand_below, below);
      __syncthreads();
      prefix = kmed;                               // the full 64-bit key
      lt_total = sel_lt + cand_lt;
      used_cand = true;
      break;
    }
  }
  const unsigned n_valid = total_cnt;
  if (n_valid > 0) {
    med_hi = __longlong_as_double((long long)prefix);
    have = true;
    // lower middle: rank n/2-1 when n is even; equals med_hi if it also lies in the final bucket
    if ((n_valid & 1u) == 0u) {
      const unsigned k1 = n_valid / 2 - 1;
      need_lower = k1 < lt_total;       // strictly-smaller elements cover rank k1 -> need max of those
      if (!need_lower) med_lo = med_hi;
    } else {
      med_lo = med_hi;
    }
  }
  if (have && need_lower && used_cand && cand_below != 0ull) {
    med_lo = __longlong_as_double((long long)cand_below);      // the lower middle lies in the same bucket
    need_lower = false;
  }
  if (have && need_lower) {
    // max over valid ratios strictly below med_hi
    double best = -1.0;
    if (worker) {
      for (int q = q_lo; q < q_hi; ++q) {
        int i, j;
        pair_ij(P, r, q, i, j);
        double ratio;
        if (!pair_ratio(cx, cy, cz, nx, ny, nz, i, j, ratio)) continue;
        if (ratio < med_hi && ratio > best) best = ratio;
      }
    }
    red[t] = best;
    __syncthreads();
    for (int s = PP_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) red[t] = red[t] > red[t + s] ? red[t] : red[t + s];
      __syncthreads();
    }
    med_lo = red[0];
    __syncthreads();
  }
  const double scale = have ? 0.5 * (med_lo + med_hi) : __builtin_nan("");

  // ---- translation = mean(cam) - mean(s*R*nocs)   (utils.py:113-118) --------------------------------
  double R[9];
  for (int i = 0; i < 9; ++i) R[i] = (double)rot[b * 9 + i];
  double sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0;
  if (t < P) {
    sx = cx[t]; sy = cy[t]; sz = cz[t];
    const double a = nx[t], bb = ny[t], c = nz[t];
    tx = (scale * R[0]) * a + (scale * R[1]) * bb + (scale * R[2]) * c;
    ty = (scale * R[3]) * a + (scale * R[4]) * bb + (scale * R[5]) * c;
    tz = (scale * R[6]) * a + (scale * R[7]) * bb + (scale * R[8]) * c;
    atomicMax((int*)&hmax[0], __float_as_int(fabsf(nx[t])));   // non-negative floats order like ints
    atomicMax((int*)&hmax[1], __float_as_int(fabsf(ny[t])));
    atomicMax((int*)&hmax[2], __float_as_int(fabsf(nz[t])));
  }
  const double mcx = block_sum(sx, red) / P, mcy = block_sum(sy, red) / P, mcz = block_sum(sz, red) / P;
  const double mtx = block_sum(tx, red) / P, mty = block_sum(ty, red) / P, mtz = block_sum(tz, red) / P;
  if (t == 0) {
    double tr[3] = {mcx - mtx, mcy - mty, mcz - mtz};
    // NaN in nocs makes np.max propagate NaN; atomicMax on bit patterns would hide it -> re-check via the sums
    bool nocs_nan = !(mtx == mtx) && (scale == scale);
    ts_out[b * 4 + 0] = tr[0]; ts_out[b * 4 + 1] = tr[1]; ts_out[b * 4 + 2] = tr[2]; ts_out[b * 4 + 3] = scale;
    // bbox corners (utils.py:49-56), sRT = [R | t] in float32 (interface_v5.py:358-361), then inv(E1)
    const double size[3] = {2.0 * (double)hmax[0] * scale, 2.0 * (double)hmax[1] * scale, 2.0 * (double)hmax[2] * scale};
    const float tf[3] = {(float)tr[0], (float)tr[1], (float)tr[2]};
    emit_bbox_world(b, R, tf, size, !nocs_nan, E1, bbox, valid);
  }
}

// One slice (blockIdx.x = pose * G + slice) of the pair passes of the split form: STAGE 0 counts the first digit, STAGE 1 re-derives
// the selected first digit from the pose's summed histogram and counts the second, STAGE 2 re-derives both and collects the keys of the
// selected 24-bit bucket (only if they fit PP_CAND, the finishing kernel's own condition).  Bucket selection is the finishing
// kernel's arithmetic on the same counts, done here by thread 0 (a 4096-bin walk: ~10 us, once or twice per workgroup).
template <int STAGE>
__global__ __launch_bounds__(PP_THREADS) void pp_split_kernel(const float* __restrict__ nocs, const float* __restrict__ depth,
                                                              const int* __restrict__ choose, const double* __restrict__ Kc,
                                                              PPScratch* __restrict__ scr, int P, int img, int G) {
  __shared__ double cx[PP_MAXP], cy[PP_MAXP], cz[PP_MAXP];
  __shared__ float nx[PP_MAXP], ny[PP_MAXP], nz[PP_MAXP];
  __shared__ unsigned hist[4096];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_rank, s_eq, s_total;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, t = threadIdx.x;
  const double fx = Kc[b * 9 + 0], fy = Kc[b * 9 + 4], pcx = Kc[b * 9 + 2], pcy = Kc[b * 9 + 5];
  if (t < P) {
    const int ch = choose[(long long)b * P + t];
    const double u = (double)(ch % img), v = (double)(ch / img);
    const double z = (double)depth[(long long)b * P + t];
    cx[t] = ((u - pcx) * z) / fx;
    cy[t] = ((v - pcy) * z) / fy;
    cz[t] = z;
    nx[t] = nocs[((long long)b * P + t) * 3 + 0];
    ny[t] = nocs[((long long)b * P + t) * 3 + 1];
    nz[t] = nocs[((long long)b * P + t) * 3 + 2];
  }
  PPScratch& S = scr[b];
  __shared__ unsigned wsum[PP_THREADS / 64];
  __shared__ unsigned f_digit, f_below, f_eq, f_total;
  // block-wide: the first bin of `hist` whose cumulative count exceeds rank rk (rk = total / 2 when half is set) — the rule of
  // postprocess_kernel's selection, 4 bins per thread + a scan over the threads' sums
  auto find_digit = [&](bool half, unsigned rk_in) {
    unsigned c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = hist[4 * t + k];
    const unsigned tsum = c[0] + c[1] + c[2] + c[3];
    unsigned incl = tsum;
    const int ln = t & 63, wv = t >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off);
      if (ln >= off) incl += o;
    }
    if (ln == 63) wsum[wv] = incl;
    if (t == 0) { f_digit = 4095u; f_below = 0u; f_eq = 0u; }
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (int k = 0; k < PP_THREADS / 64; ++k) { const unsigned w = wsum[k]; if (k < wv) wbase += w; total += w; }
    const unsigned excl = wbase + incl - tsum;
    const unsigned rk = half ? total / 2 : rk_in;
    if (t == 0) f_total = total;
    if (rk >= excl && rk < excl + tsum) {
      unsigned acc = excl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (acc + c[k] > rk) { f_digit = (unsigned)(4 * t + k); f_below = acc; f_eq = c[k]; break; }
        acc += c[k];
      }
    }
    __syncthreads();
  };
  if (STAGE >= 1) {
    for (int i = t; i < 4096; i += PP_THREADS) hist[i] = S.hist0[i];
    __syncthreads();
    find_digit(true, 0u);
    unsigned long long pf = (unsigned long long)f_digit;
    const unsigned total = f_total;
    unsigned rk = total / 2 - f_below, eq = f_eq;
    __syncthreads();
    if (STAGE >= 2 && total > 0) {
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = S.hist1[i];
      __syncthreads();
      find_digit(false, rk);
      pf = (pf << 12) | (unsigned long long)f_digit;
      eq = f_eq;
      __syncthreads();
    }
    if (t == 0) { s_prefix = pf; s_rank = rk; s_eq = eq; s_total = total; }
    __syncthreads();
  }
  for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
  __syncthreads();
  if (STAGE >= 1 && s_total == 0) return;
  if (STAGE == 2 && s_eq > (unsigned)PP_CAND) return;        // the finishing kernel will run the remaining radix passes itself
  const unsigned long long prefix = STAGE >= 1 ? s_prefix : 0ull;
  // the pairs of this slice: row pair r, a 1 / (parts * G) share of its P - 1 partners (postprocess_kernel's map, cut G times finer)
  const int npair_rows = P / 2, per_row = P - 1;
  const int parts = (PP_THREADS / npair_rows) * G;
  const int r = t % npair_rows, part = (t / npair_rows) * G + g;
  if (t / npair_rows < PP_THREADS / npair_rows) {
    const int q_lo = (int)(((long long)per_row * part) / parts), q_hi = (int)(((long long)per_row * (part + 1)) / parts);
    for (int q = q_lo; q < q_hi; ++q) {
      int i, j;
      pair_ij(P, r, q, i, j);
      double ratio;
      if (!pair_ratio(cx, cy, cz, nx, ny, nz, i, j, ratio)) continue;
      const unsigned long long key = (unsigned long long)__double_as_longlong(ratio);
      if (STAGE == 0) atomicAdd(&hist[(unsigned)(key >> 52) & 4095u], 1u);
      else if (STAGE == 1) { if ((key >> 52) == prefix) atomicAdd(&hist[(unsigned)(key >> 40) & 4095u], 1u); }
      else if ((key >> 40) == prefix) S.cand[atomicAdd(&S.ncand, 1u)] = key;
    }
  }
  if (STAGE < 2) {
    __syncthreads();
    unsigned* gh = STAGE == 0 ? S.hist0 : S.hist1;
    for (int i = t; i < 4096; i += PP_THREADS) { const unsigned c = hist[i]; if (c) atomicAdd(&gh[i], c); }
  }
}

size_t postprocess_scratch_bytes(int B) { return (size_t)B * sizeof(PPScratch); }

// slices per pose of the split form: fill the chip (256 CUs) without exceeding 32 slices; 1 = the one-kernel form
int postprocess_slices(int B) {
  int g = 1;
  while (g < 32 && (long long)B * g * 2 <= 256) g *= 2;
  return g;
}

int launch_postprocess(const float* nocs, const float* depth, const float* rot, const int* choose, const double* Kc,
                       const double* E1, double* bbox, double* ts_out, int* valid, int B, int P, int img, hipStream_t s,
                       void* scratch, size_t scratch_bytes) {
  RGBM_REQUIRE(P >= 2 && P <= PP_MAXP && (P % 2) == 0 && (PP_THREADS % (P / 2)) == 0, "postprocess needs even P<=1024 dividing 2048");
  const int G = scratch ? postprocess_slices(B) : 1;
  if (G > 1 && !(g_debug_flags & (1 << 23))) {        // debug flag 8388608: one-kernel form even when scratch is given (A/B)
    RGBM_REQUIRE(scratch_bytes >= postprocess_scratch_bytes(B) && ((uintptr_t)scratch & 7) == 0, "postprocess scratch too small or misaligned");
    PPScratch* scr = reinterpret_cast<PPScratch*>(scratch);
    RGBM_CHECK_HIP(hipMemsetAsync(scr, 0, postprocess_scratch_bytes(B), s));
    hipLaunchKernelGGL(pp_split_kernel<0>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G);
    hipLaunchKernelGGL(pp_split_kernel<1>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G);
    hipLaunchKernelGGL(pp_split_kernel<2>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G);
    hipLaunchKernelGGL(postprocess_kernel<true>, dim3(B), dim3(PP_THREADS), 0, s, nocs, depth, rot, choose, Kc, E1, bbox, ts_out,
                       valid, P, img, (const PPScratch*)scr);
  } else {
    hipLaunchKernelGGL(postprocess_kernel<false>, dim3(B), dim3(PP_THREADS), 0, s, nocs, depth, rot, choose, Kc, E1, bbox, ts_out,
                       valid, P, img, (const PPScratch*)nullptr);
  }
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
