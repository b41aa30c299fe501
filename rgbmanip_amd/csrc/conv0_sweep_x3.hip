// conv0 of the cost-regularisation net + plane sweep for the BF16X3 (split pair) mode: the depth-sweeping producer / consumer
// structure of conv0_sweep.hip (Conv3d 32->8 k3 p1 + BN + ReLU over "ref + warped", network_v5.py:260-291,378-430) with
//   * fp32 arithmetic up to the MFMA operands: the feature map is read as plain fp32 (a copy the forward makes once), the
//     bilinear blend runs in fp32, and only then each voxel is split into bf16 hi + lo (common.h, bx3_t) — so the volume
//     the convolution sees carries 16 significand bits (a single-term 16-bit volume alone puts the depth output at 2.4e-4
//     from the reference, tools/split_emulation.py);
//   * three products per tap: W_lo*x_hi, W_hi*x_lo, W_hi*x_hi on v_mfma_f32_16x16x32_bf16, fp32 accumulation - since round 6 as FIVE
//     MFMAs per (in-plane tap, fragment) instead of six: the 16 MFMA rows are 8 output channels x two row groups, and the nine
//     (depth tap, product) row groups of a voxel pair up as  [W_lo(kd0); W_lo(kd1)] x_hi,  [W_hi(kd0); W_hi(kd1)] x_lo,
//     [W_hi(kd0); W_hi(kd1)] x_hi  (accumulator Xn, as before)  and  [W_lo(kd2); W_hi(kd2)] x_hi,  [W_hi(kd2); 0] x_lo  (accumulator Y:
//     rows 0-7 collect the two small products, rows 8-15 the large one; summed in the epilogue) - before, the three kd = 2 products each
//     filled half an instruction;
//   * one persistent workgroup per CU walking its tiles (the 121 KB plane ring and the 36 weight operands a consumer lane
//     keeps in registers allow one resident workgroup anyway): weights are loaded once per launch, the plane ring runs on
//     across tile boundaries.
// LDS voxel = [hi: 32 channels, 64 B][lo: 64 B][32 B pad]: with the 160-byte stride the 16 consecutive voxels x 4 k-groups a
// ds_read_b128 lane group touches fall on distinct banks (the 80-byte voxels of the 16-bit kernel are 2-way conflicted).
#include <string.h>

#include "common.h"
#include "kernels.h"
#include "prof.h"

#ifndef X3_DEPTH
#define X3_DEPTH 4    // cooperative producers: rounds of gathers in flight (half a plane ahead)
#endif

namespace rgbm {

namespace {

constexpr int X3_TH = 12, X3_TW = 16;
constexpr int X3_HH = X3_TH + 2, X3_HW = X3_TW + 2;
constexpr int X3_NV = X3_HH * X3_HW;             // 252 voxels per input plane incl. halo
constexpr int X3_VS = 160;
constexpr int X3_SLOT = X3_NV * X3_VS;           // 40320
constexpr int X3_NSLOT = 3;                      // plane p is written during step p, read during step p + 1, rewritten during step p + 3
constexpr int X3_RING = X3_NSLOT * X3_SLOT;      // 120960
constexpr int X3_WLDS = 9 * 64 * 16;             // lo operands of the kd=2 taps, lane-linear per tap (the consumers' register budget)
constexpr int X3_REC = 4 * 2 * 64 * 32;   // cooperative producers: per wave [plane parity][lane] {4 corner offsets, 4 weights}
constexpr int X3_LDS = X3_RING + X3_WLDS + X3_REC;
constexpr int X3_NPW = 4, X3_NCW = 4;            // producer waves and consumer waves, one of each per SIMD (round 6; rounds 2-5: three consumer waves of 4 rows)
constexpr int X3_CR = X3_TH / X3_NCW;            // rows of 16 voxels (= fragments) per consumer wave: 3
static_assert(X3_CR == 3, "the consumers' unit schedule below is written for three fragments per wave");
constexpr int X3_THREADS = (X3_NPW + X3_NCW) * 64;
static_assert(X3_NV <= X3_NPW * 64, "one producer thread per voxel");

struct Sweep3Desc {
  const float* feat;        // [V][H][W][32] fp32
  const uint4* wgt;         // [2][18][16 rows][4 k-groups] bf16x8: hi operands (A01[9], A2[9]) then the lo operands
  const float* bias;        // [16] folded BN shift
  const float* homog;       // [V][12]
  const float* depths;      // [B][D]
  bx3_t* out;               // [N][D][H][W][8]
  int N, D, H, W, v0, V, B, nth, ntw, relu, n_tiles;
  const int* tile_list; const int* tile_count;      // sparse cost regularisation: only these tiles (ascending), else null
};

typedef __attribute__((ext_vector_type(4))) float f4v;        // native vectors: usable as tied inline-asm operands
typedef __attribute__((ext_vector_type(4))) unsigned u4v;
__device__ __forceinline__ u4v ld_u4v(const uint4* p) { const uint4 t = *p; return u4v{t.x, t.y, t.z, t.w}; }

__device__ __forceinline__ f32x4 mma32(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 4 channels of one voxel: ref + w0*a + w1*b + w2*c + w3*e as one fma chain (whole-vector arithmetic: the gathered registers
// are tied inline-asm results, see hazard (4) in DESIGN.md about indexing them with an unrolled loop counter)
__device__ __forceinline__ f4v blend4(const f4v& r, const f4v& a, const f4v& b, const f4v& c, const f4v& e, const float* w) {
  f4v t = a * w[0] + r;
  t = b * w[1] + t;
  t = c * w[2] + t;
  t = e * w[3] + t;
  return t;
}

// 8 fp32 channels -> their bf16 hi parts (16 bytes) and lo parts (16 bytes)
__device__ __forceinline__ void split8(const float* o, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = pack2_bf16(o[2 * q], o[2 * q + 1]);
    l[q] = pack2_bf16(o[2 * q] - __uint_as_float(h[q] << 16), o[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace

// one workgroup per CU (<= 256 VGPRs): 8 waves, a producer and a consumer per SIMD
__global__ __launch_bounds__(X3_THREADS, 2) void conv0_sweep_x3_kernel(const Sweep3Desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = d.D, H = d.H, W = d.W;
  const int n_tiles = d.tile_list ? d.tile_count[0] : d.n_tiles;      // wave-uniform: every role walks the same tiles
  const int n_my = n_tiles > (int)blockIdx.x ? (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  // XCD-aware tile order (the grid is a multiple of 8): every XCD walks a contiguous run of tiles, i.e. whole views, so the
  // partner feature maps its CUs gather from stay in that XCD's L2
  auto tile_of = [&](int k, int& n, int& h0, int& w0) {
    const int v = (int)blockIdx.x + k * (int)gridDim.x;
    const int nblk = n_tiles, bq = nblk >> 3, br = nblk & 7, xcd = v & 7, bidx = v >> 3;
    int t = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
    if (d.tile_list) t = d.tile_list[t];
    const int tw = t % d.ntw; t /= d.ntw;
    const int th = t % d.nth; t /= d.nth;
    n = t; h0 = th * X3_TH; w0 = tw * X3_TW;
  };

  // the [W_hi(kd=2); 0] operands of the nine in-plane taps -> LDS (read by the consumers every plane): lane-linear rows [tap][lane].
  // d.wgt holds conv0_sweep_pack's A2 block as rows 0-7 = 0, rows 8-15 = W(kd=2): MFMA row r < 8 of this operand is its row r + 8.
  for (int i = tid; i < 9 * 64; i += X3_THREADS) {
    const int s9 = i >> 6, ln = i & 63, row = ln & 15;
    reinterpret_cast<uint4*>(planes + X3_RING)[i] = row < 8 ? d.wgt[((9 + s9) * 16 + row + 8) * 4 + (ln >> 4)] : make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  if (wave < X3_NPW) {
    // ------------------------------------------------------------------ producers, cooperative gathers, streaming across tiles
    // Thread pv OWNS voxel pv of the 14 x 18 plane for the projection (corner offsets + bilinear weights), but the gathers, the
    // blend and the LDS store are done per (voxel, 16-byte chunk): in round r lane (j = lane >> 3, c = lane & 7) works on chunk c
    // (4 fp32 channels) of voxel wave*64 + r*8 + j, so the eight lanes of a group read the eight chunks of one pixel — one 128-byte
    // line — and a gather instruction touches 8 lines instead of 64.  Measured on the first version (one lane = one voxel = all its
    // chunks; removed in round 6): the vector L1 looks up ~1.3 lines per clock and CU under these gathers, and 32 instructions x 64
    // lines x 4 waves per plane were the kernel's longest pole (41.5 ms; 33.9 ms when groups of 8 lanes were pointed at the same
    // pixel).  The owner publishes its record {off[4], w[4]} per plane in LDS (wave-private, two parities); LDS executes a wave's
    // instructions in order, so the readers of the same wave need no barrier.
    //
    // Round 6: the plane stream runs on ACROSS tiles.  Rounds 1-5 drained the gathers at a tile's end, sat at a "step D" barrier while
    // the consumers multiplied the tile's last plane, and only then loaded the next tile's reference features and homography,
    // projected its first plane and requested its gathers.  Worth -1 % (17.72 -> 17.50 ms on the survey masks, 29.55 -> 29.25 dense,
    // same box): the D-independent 8 % of a launch that tools/sweep_depth_fit.py measures is still there afterwards, i.e. it is not
    // the hand-over (a tile's first touches of its partner's feature rows are the likelier cost).  Kept for what it removes — the drain,
    // a barrier per tile, two producer variants.  A tile's setup loads are requested a whole tile ahead (tile_request), the LAST plane of a tile requests plane 0 of
    // the next one where the planes before it request their successor (from the partner view of the next tile: the gathers' base is a
    // macro argument), and there is no barrier but the one per plane: global plane q is produced during step q into ring slot q % 3 and
    // multiplied during step q + 1.  The last plane is peeled out of the plane loop, so the loop with the counted waits keeps a single
    // back edge (tools/check_asm_gathers.py follows the rolling gather registers to the end of the kernel).
    static_assert(X3_DEPTH == 4, "four rounds of gathers in flight (half a plane ahead)");
    const int pv = tid;
    const bool act = pv < X3_NV;
    const int hh = pv / X3_HW, hw = pv - hh * X3_HW;
    const int j = lane >> 3, c = lane & 7;
    unsigned char* rec = planes + X3_RING + X3_WLDS + wave * (2 * 64 * 32);
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    // [round & 3][corner]: the data of a round is requested four rounds (half a plane) ahead, 16 gathers in flight per lane.  Four
    // sets, not eight: with 128 registers of in-flight loads the register allocator split their live ranges around the
    // projection code (v_mov copies of registers whose loads had not landed, and the freed registers reused as temporaries under
    // the landing loads: wild gather offsets, a memory fault on the first test shape) — an inline-asm load is final for the
    // compiler the moment it is issued, so the only protection is to leave it no reason to move such a register.
    f4v g[X3_DEPTH][4];
#pragma unroll
    for (int r = 0; r < X3_DEPTH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[r][q] = f4v{0.f, 0.f, 0.f, 0.f};
    // voxel of round r for this lane: wave*64 + r*8 + j; byte position of chunk c's hi part inside a plane slot (lo part 64 bytes on)
    const int vdst0 = (wave * 64 + j) * X3_VS + c * 8;
    const bool vact7 = wave * 64 + 56 + j < X3_NV;      // only round 7 of the last producer wave runs past the 252 voxels
#define X3_GATHER(R, Q, OFF, BASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(g[R][Q]) : "v"(OFF), "s"(BASE) : "memory")
#define X3_GATHER4(R, O, BASE) do { X3_GATHER(R, 0, (O).x, BASE); X3_GATHER(R, 1, (O).y, BASE); X3_GATHER(R, 2, (O).z, BASE); X3_GATHER(R, 3, (O).w, BASE); } while (0)
#define X3_WAITB(R) asm volatile("s_waitcnt vmcnt(12)" : "+v"(g[R][0]), "+v"(g[R][1]), "+v"(g[R][2]), "+v"(g[R][3]) :: "memory")
    // what the projection of a tile's planes needs, per lane = per owned voxel
    struct TileP { float rx, ry, rz, t0, t1, t2; int dbits; int inb; };      // (inb an int: a bool member made hipcc copy the struct's tail through scratch)
    // a tile's setup loads in flight (nothing derived yet): lane e < 12 holds word e of the view's homography, lane z < D depth z
    struct TileRaw { float hmv; int dbits; int gh, gw; };
    auto tile_request = [&](int k, TileRaw& R, f4v (&rf)[8], const unsigned char*& base) {
      int n, h0, w0;
      tile_of(k, n, h0, w0);
      const int vv = d.v0 + n;
      const int partner = (vv + d.B) % d.V, bb = vv % d.B;
      R.gh = h0 - 1 + hh; R.gw = w0 - 1 + hw;
      R.hmv = lane < 12 ? d.homog[(long long)vv * 12 + lane] : 0.f;
      R.dbits = __float_as_int(lane < D ? d.depths[(long long)bb * D + lane] : 1.f);      // lane z holds depth z
      base = reinterpret_cast<const unsigned char*>(d.feat + (long long)partner * H * W * 32);
      // reference-view features of this lane's eight (voxel, chunk) pairs; outside the image: conv zero padding
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int pvx = wave * 64 + r * 8 + j;
        const int hr = pvx / X3_HW, wr = pvx - hr * X3_HW;
        const int ghr = h0 - 1 + hr, gwr = w0 - 1 + wr;
        rf[r] = f4v{0.f, 0.f, 0.f, 0.f};
        if (pvx < X3_NV && (unsigned)ghr < (unsigned)H && (unsigned)gwr < (unsigned)W)
          rf[r] = *reinterpret_cast<const f4v*>(d.feat + (((long long)vv * H + ghr) * W + gwr) * 32 + c * 4);
      }
    };
    auto tile_finish = [&](const TileRaw& R, TileP& P) {
      float hm[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) hm[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(R.hmv), e));
      // projection of the OWNED pixel (network_v5.py:378-430 folded as in conv0_sweep.hip): p = rot*(x,y,1)*depth + t
      const float x = (float)R.gw, y = (float)R.gh;
      P.rx = hm[0] * x + hm[1] * y + hm[2];
      P.ry = hm[3] * x + hm[4] * y + hm[5];
      P.rz = hm[6] * x + hm[7] * y + hm[8];
      P.t0 = hm[9]; P.t1 = hm[10]; P.t2 = hm[11];
      P.dbits = R.dbits;
      P.inb = (act && (unsigned)R.gh < (unsigned)H && (unsigned)R.gw < (unsigned)W) ? 1 : 0;
    };
    // corner record of the owned voxel at depth plane z of the tile described by P -> LDS record [parity][lane]
    auto publish = [&](int z, int par, const TileP& P) {
      const float depth = __int_as_float(__builtin_amdgcn_readlane(P.dbits, z));
      const float px = P.rx * depth + P.t0, py = P.ry * depth + P.t1, pz = P.rz * depth + P.t2;
      const float ix = (px / pz) * sx - 0.5f, iy = (py / pz) * sy - 0.5f;       // true divisions: the fp32 path's accuracy
      const bool fin = isfinite(ix) && isfinite(iy);
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;                  // v_cvt_i32_f32 saturates: far-away projections stay "outside"
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)(x0 + 1) < (unsigned)W;
      const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)(y0 + 1) < (unsigned)H;
      const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0, -1) + 1, W - 1);
      const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0, -1) + 1, H - 1);
      const float ux = 1.f - tx, uy = 1.f - ty;
      f4v w;
      w.x = (P.inb && xin0 && yin0) ? ux * uy : 0.f;
      w.y = (P.inb && xin1 && yin0) ? tx * uy : 0.f;
      w.z = (P.inb && xin0 && yin1) ? ux * ty : 0.f;
      w.w = (P.inb && xin1 && yin1) ? tx * ty : 0.f;
      if (P.inb && !fin) w.x = __builtin_nanf("");           // the voxel becomes NaN like the reference's
      const unsigned r0 = (unsigned)(yc0 * W), r1 = (unsigned)(yc1 * W);
      u4v off;
      off.x = (r0 + (unsigned)xc0) * 128u;
      off.y = (r0 + (unsigned)xc1) * 128u;
      off.z = (r1 + (unsigned)xc0) * 128u;
      off.w = (r1 + (unsigned)xc1) * 128u;
      unsigned char* p = rec + par * 2048 + lane * 32;
      *reinterpret_cast<u4v*>(p) = off;
      *reinterpret_cast<f4v*>(p + 16) = w;
    };
    auto rec_off = [&](int par, int r) {       // corner offsets of round r's voxel, already at this lane's chunk
      u4v o = *reinterpret_cast<const u4v*>(rec + par * 2048 + (r * 8 + j) * 32);
      const unsigned cc = (unsigned)c * 16u;
      o.x += cc; o.y += cc; o.z += cc; o.w += cc;
      return o;
    };
    auto rec_w = [&](int par, int r) { return *reinterpret_cast<const f4v*>(rec + par * 2048 + (r * 8 + j) * 32 + 16); };
    // round R of the plane whose gathers are in flight: wait for its 4 gathers (the oldest), blend chunk c of its voxel, request the
    // data of the round four rounds on — this plane's round R + 4 (R < 4: from BASE, this tile's partner view), or round R - 4 of the
    // plane that follows in the stream (R >= 4: from the partner view of THAT plane's tile) — into the same registers, split into bf16
    // hi / lo and store 8 + 8 bytes.  The records of the next round are read under the split.
#define X3_ROUND(R, BASE)                                                                                        \
    {                                                                                                            \
      X3_WAITB((R) % X3_DEPTH);                                                                                  \
      const float wq[4] = {wc.x, wc.y, wc.z, wc.w};                                                              \
      f4v t = blend4(ref[R], g[(R) % X3_DEPTH][0], g[(R) % X3_DEPTH][1], g[(R) % X3_DEPTH][2], g[(R) % X3_DEPTH][3], wq); \
      /* the blend must be DONE before the registers are requested again: without this tie hipcc sank round 7's blend into \
         the `vact7` branch below the gathers and kept "the old values" in copies made before the wait, i.e. stale ones */ \
      asm volatile("" : "+v"(t));                                                                                \
      X3_GATHER4((R) % X3_DEPTH, oc, BASE);                                                                      \
      if (R < 7) { wc = rec_w(pc, R + 1);                                                                        \
                   oc = (R + 1 < 4) ? rec_off(pc, R + 5) : rec_off(pn, R - 3); }                                 \
      const unsigned h0_ = pack2_bf16(t.x, t.y), h1_ = pack2_bf16(t.z, t.w);                                     \
      const unsigned l0_ = pack2_bf16(t.x - __uint_as_float(h0_ << 16), t.y - __uint_as_float(h0_ & 0xffff0000u)); \
      const unsigned l1_ = pack2_bf16(t.z - __uint_as_float(h1_ << 16), t.w - __uint_as_float(h1_ & 0xffff0000u)); \
      if (R < 7 || vact7) {                                                                                      \
        *reinterpret_cast<uint2*>(dst + vdst0 + R * 8 * X3_VS) = make_uint2(h0_, h1_);                           \
        *reinterpret_cast<uint2*>(dst + vdst0 + R * 8 * X3_VS + 64) = make_uint2(l0_, l1_);                      \
      }                                                                                                          \
    }
    // one plane: publish the record of the plane that follows it in the stream (depth ZREQ of the tile PREQ), blend and store this one,
    // request that one (rounds 0-3 of it from BASEN), meet the consumers
#define X3_PLANE(PREQ, ZREQ, BASEC, BASEN)                                                                       \
    {                                                                                                            \
      const int pc = par, pn = par ^ 1;                                                                          \
      publish((ZREQ), pn, (PREQ));                                                                               \
      unsigned char* dst = planes + (pg % X3_NSLOT) * X3_SLOT;                                                   \
      f4v wc = rec_w(pc, 0);                                                                                     \
      u4v oc = rec_off(pc, 4);                                                                                   \
      X3_ROUND(0, BASEC) X3_ROUND(1, BASEC) X3_ROUND(2, BASEC) X3_ROUND(3, BASEC)                                \
      X3_ROUND(4, BASEN) X3_ROUND(5, BASEN) X3_ROUND(6, BASEN) X3_ROUND(7, BASEN)                                \
      ++pg;                                                                                                      \
      par ^= 1;                                                                                                  \
      /* the plane just written must be visible before the consumers are released; prefetched gathers stay in flight */ \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
      __builtin_amdgcn_s_barrier();                                                                              \
      asm volatile("" ::: "memory");          /* no LDS store of the next plane may be scheduled above the barrier */ \
    }
    int pg = 0, par = 0;                                 // planes produced so far (ring slot = pg % 3), parity of the plane's record
    if (n_my > 0) {
      TileRaw raw;
      TileP cur, nxt;
      f4v ref[8], nref[8];
      const unsigned char *srcb, *nsrcb;
      tile_request(0, raw, ref, srcb);
      tile_finish(raw, cur);
      publish(0, 0, cur);
      {
        u4v o;
        o = rec_off(0, 0); X3_GATHER4(0, o, srcb);
        o = rec_off(0, 1); X3_GATHER4(1, o, srcb);
        o = rec_off(0, 2); X3_GATHER4(2, o, srcb);
        o = rec_off(0, 3); X3_GATHER4(3, o, srcb);
      }
      nsrcb = srcb;
      for (int k = 0; k < n_my; ++k) {
        const bool has_next = k + 1 < n_my;
        // the next tile's setup loads travel while this tile's planes are produced; their first use is behind the plane loop (they are
        // OLDER than every gather issued after them, so the counted waits below can only wait longer, never shorter)
        if (has_next) tile_request(k + 1, raw, nref, nsrcb);
        for (int z = 0; z + 1 < D; ++z) {
#pragma unroll
          for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(ref[r]));      // (pinned per plane: nothing of the reference features is hoisted around the counted gathers)
          X3_PLANE(cur, z + 1, srcb, srcb)
        }
        // last plane of the tile: what is requested behind it is plane 0 of the next tile (none left: a harmless re-request of this
        // plane keeps the counted waits valid)
        if (has_next) tile_finish(raw, nxt); else nxt = cur;
        X3_PLANE(nxt, has_next ? 0 : D - 1, srcb, nsrcb)
        if (has_next) {
          cur = nxt;
          srcb = nsrcb;
#pragma unroll
          for (int r = 0; r < 8; ++r) ref[r] = nref[r];
        }
      }
      __builtin_amdgcn_s_barrier();                        // the consumers multiply the stream's last plane
      asm volatile("" ::: "memory");
    }
#undef X3_GATHER
#undef X3_GATHER4
#undef X3_WAITB
#undef X3_ROUND
#undef X3_PLANE
  } else {
    // ------------------------------------------------------------------ consumers: 3 rows x 16 voxels per wave
    const int cw = wave - X3_NPW;
    const int lr = lane & 15, lg = lane >> 4;
    // Weights of all 9 in-plane taps (conv0_sweep_pack's paired-tap scheme, d.wgt = [hi: A01[9], A2[9]][lo: A01[9], A2[9]] with
    // A01 rows 0-7 = W(kd=0), rows 8-15 = W(kd=1) and A2 rows 0-7 = 0, rows 8-15 = W(kd=2)):
    //   A01h, A01l : hi / lo of [W(kd0); W(kd1)]
    //   A2x        : rows 0-7 = lo of W(kd2), rows 8-15 = hi of W(kd2)     (regrouped from the A2 blocks at load time)
    //   AL (LDS)   : rows 0-7 = hi of W(kd2), rows 8-15 = 0                (one lane-linear 1 KB row per tap, see above)
    // 108 registers, loaded once per launch; with the fourth set in registers too the kernel spilled at the 256 cap (8 waves on 4 SIMDs:
    // two waves share a SIMD's 512 registers), and a spill reload in a consumer waits on vmcnt, i.e. on its output stores.
    u4v A01h[9], A01l[9], A2x[9];
    // Operand registers of the plane loop.  The first version of this loop was plain C++ (ds_read into `uint4` temporaries,
    // MFMA builtins, hipcc's own waits): about one (workgroup, consumer wave, input plane) in a thousand came out wrong,
    // differently from run to run, always as ONE consumer wave's rows of three consecutive output planes — the signature of
    // hazard (3) of the 16-bit sweep in DESIGN.md, where a third ring slot only moved the timing.  Established on the device:
    // it is not the ring protocol (two barriers per plane: same rate), not the counted gathers (vmcnt(0) before every blend:
    // same rate), not the ring's contents (a build that dumps what the consumers READ from the ring — first, centre and
    // last tap — never showed a wrong value); 64 wait states of s_nop between a step's MFMAs and the next step's ds_reads
    // made every run exact, 32 did not.  hipcc's schedule issues the next operands' ds_reads into registers that MFMAs issued
    // a few instructions earlier still name as sources or accumulator inputs (it rotates accumulators, vDst != SrcC, and
    // pads such reuses with `s_nop 2`).  A stand-alone reproduction of exactly those register reuses
    // (tools/micro/mfma_lds_war.hip: B-operand reload, accumulator-input reload, dependent chains, with and without a
    // VALU / ds_write stress wave on the same SIMD) shows NO corruption, so the mechanism is not isolated; what is established
    // is the cure: nothing is left to the register allocator or the scheduler here.  Accumulators are updated in place by
    // inline-asm MFMAs and never serve as load destinations, operands rotate through SIX fixed register sets (a set is
    // reloaded only after three further units = 15 MFMAs have been issued behind its last reader), LDS reads and their counted
    // waits are placed by hand.  Bit-stable over repeated runs at every size tried
    // (test_conv0_sweep_bf16x3_matches_volume_then_conv, tools/debug_sweep_x3.py).
    u4v B[6][2];                       // [set][hi, lo] of one (tap, fragment) unit
    u4v AL[3];                         // the [W_hi(kd2); 0] operand (from LDS), one per tap in flight
#pragma unroll
    for (int i = 0; i < 3; ++i) AL[i] = u4v{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 6; ++i) B[i][0] = B[i][1] = u4v{0u, 0u, 0u, 0u};
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)planes;
    const unsigned wa = lds_base + X3_RING + lane * 16;      // this lane's LDS-resident operand of tap 0
#define X3_BOFF(TP, F) ((((F) + (TP) / 3) * X3_HW + (TP) % 3) * X3_VS)
    // unit U = 3 * tap + fragment: its hi / lo operand pair into set U % 6
#define X3_LDB(U)                                                                                                    \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                    \
                 : "+v"(B[(U) % 6][0]), "+v"(B[(U) % 6][1])                                                           \
                 : "v"(sa), "n"(X3_BOFF((U) / 3, (U) % 3)), "n"(X3_BOFF((U) / 3, (U) % 3) + 64) : "memory");
#define X3_LDW(TP) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(AL[(TP) % 3]) : "v"(wa), "n"((TP) * 1024) : "memory");
#define X3_MFMA(ACC, A, BB) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(BB))
    // the five MFMAs of a unit: small products first (as in rounds 2-5), the two accumulators alternating
#define X3_MMA5(U)                                                                                                   \
    X3_MFMA(Xn[(U) % 3], A01l[(U) / 3], B[(U) % 6][0]);                                                              \
    X3_MFMA(Y[(U) % 3], A2x[(U) / 3], B[(U) % 6][0]);                                                                \
    X3_MFMA(Xn[(U) % 3], A01h[(U) / 3], B[(U) % 6][1]);                                                              \
    X3_MFMA(Y[(U) % 3], AL[((U) / 3) % 3], B[(U) % 6][1]);                                                           \
    X3_MFMA(Xn[(U) % 3], A01h[(U) / 3], B[(U) % 6][0]);
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
      A01h[s9] = ld_u4v(d.wgt + (s9 * 16 + lr) * 4 + lg);
      A01l[s9] = ld_u4v(d.wgt + ((18 + s9) * 16 + lr) * 4 + lg);
      // rows 0-7: lo of W(kd2) = row lr + 8 of the lo A2 block; rows 8-15: hi of W(kd2) = row lr of the hi A2 block
      A2x[s9] = lr < 8 ? ld_u4v(d.wgt + ((27 + s9) * 16 + lr + 8) * 4 + lg) : ld_u4v(d.wgt + ((9 + s9) * 16 + lr) * 4 + lg);
    }
    const int ch = (lg & 1) * 4;       // after the lane-half swap: fragment (lg < 2 ? first : second of the pair), voxel lr, channels ch..ch+3
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = d.bias[ch + r];
    const int boff = ((cw * X3_CR) * X3_HW + lr) * X3_VS + lg * 16;     // fragment 0, tap (0,0), hi part
    int pg0 = 0;                                                     // ring index of this tile's plane 0
    for (int k = 0; k < n_my; ++k) {
      int n, h0, w0;
      tile_of(k, n, h0, w0);
      const int ow = w0 + lr;
      // output rows of the two fragment pairs of the epilogue: (0, 1) and (2, 2) — the third fragment pairs with itself, its copy in
      // the upper lane half is not stored
      int oh[2];
      bool ook[2];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const bool self = pr == 1;
        oh[pr] = h0 + cw * X3_CR + pr * 2 + (self ? 0 : (lg >> 1));
        ook[pr] = oh[pr] < H && ow < W && !(self && lg >= 2);
      }
      f32x4 Xp[3], Xn[3], Y[3], Lp[2];
#pragma unroll
      for (int f = 0; f < 3; ++f) Xp[f] = Xn[f] = Y[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      Lp[0] = Lp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      // out plane o = rows 8-15 of Xp (kd=1 taps of plane o) + rows 0-7 of X[o-1] (kd=0 taps of plane o-1, kept in Lp) + the kd=2 taps of
      // plane o+1, which the plane just multiplied left in Y (rows 8-15: W_hi x_hi, rows 0-7: W_lo x_hi + W_hi x_lo); leaves Lp = rows 0-7 of Xp
      auto emit = [&](int o) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int f0 = 2 * pr, f1 = pr == 0 ? 1 : 2;
          f32x4 hi, lo, yh, yl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Xp[f0][r]), __float_as_uint(Xp[f1][r]), false, false);
            lo[r] = __uint_as_float(sw[0]);
            hi[r] = __uint_as_float(sw[1]);
            const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(Y[f0][r]), __float_as_uint(Y[f1][r]), false, false);
            yl[r] = __uint_as_float(sy[0]);
            yh[r] = __uint_as_float(sy[1]);
          }
          if (o >= 0 && ook[pr]) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = ((hi[r] + Lp[pr][r]) + (yh[r] + yl[r])) + bias[r];
              if (d.relu) v[r] = v[r] < 0.f ? 0.f : v[r];            // NaN propagates, like torch.relu
            }
            store4(d.out + ((((long long)n * D + o) * H + oh[pr]) * W + ow) * 8 + ch, v);
          }
          Lp[pr] = lo;
        }
      };
      // (k > 0: plane 0 of this tile was produced, and its barrier passed, while the previous tile's last plane was multiplied)
      for (int z = k == 0 ? 0 : 1; z <= D; ++z) {
        if (z >= 1) {
          const int p = z - 1;
          const unsigned sa = lds_base + (unsigned)(((pg0 + p) % X3_NSLOT) * X3_SLOT + boff);      // LDS address of (fragment 0, tap (0,0), hi)
#pragma unroll
          for (int f = 0; f < 3; ++f) Xn[f] = Y[f] = f32x4{0.f, 0.f, 0.f, 0.f};
          // (the zeroing happens HERE, not as v_mov instructions the compiler drops between the inline-asm MFMAs that name these
          // registers as accumulator inputs: it cannot see that those statements are MFMAs and owes them no wait states)
          asm volatile("s_nop 1" : "+v"(Xn[0]), "+v"(Xn[1]), "+v"(Xn[2]), "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]));
          // 27 units per plane, unit u = (in-plane tap u / 3, fragment u % 3): two operand reads (three with the tap's LDS-resident
          // weights, which ride with the tap's first unit) and five MFMAs.  The reads of unit u + 3 are issued behind the MFMAs of
          // unit u, into the register set unit u - 3 multiplied from.  Counted waits: at unit u the reads of units u .. u + 2 are
          // outstanding, LDS operations complete in order, so unit u's have landed once at most those of u + 1 and u + 2 remain:
          // 4, or 5 when one of them opens a tap.
          X3_LDW(0) X3_LDB(0) X3_LDB(1) X3_LDB(2)
#define X3_UNIT(U, NEXT, WAITN)                                                                                      \
          asm volatile("s_waitcnt lgkmcnt(" #WAITN ")" ::: "memory");                                                \
          X3_MMA5(U)                                                                                                 \
          NEXT
          X3_UNIT(0, X3_LDW(1) X3_LDB(3), 4)    X3_UNIT(1, X3_LDB(4), 5)    X3_UNIT(2, X3_LDB(5), 5)
          X3_UNIT(3, X3_LDW(2) X3_LDB(6), 4)    X3_UNIT(4, X3_LDB(7), 5)    X3_UNIT(5, X3_LDB(8), 5)
          X3_UNIT(6, X3_LDW(3) X3_LDB(9), 4)    X3_UNIT(7, X3_LDB(10), 5)   X3_UNIT(8, X3_LDB(11), 5)
          X3_UNIT(9, X3_LDW(4) X3_LDB(12), 4)   X3_UNIT(10, X3_LDB(13), 5)  X3_UNIT(11, X3_LDB(14), 5)
          X3_UNIT(12, X3_LDW(5) X3_LDB(15), 4)  X3_UNIT(13, X3_LDB(16), 5)  X3_UNIT(14, X3_LDB(17), 5)
          X3_UNIT(15, X3_LDW(6) X3_LDB(18), 4)  X3_UNIT(16, X3_LDB(19), 5)  X3_UNIT(17, X3_LDB(20), 5)
          X3_UNIT(18, X3_LDW(7) X3_LDB(21), 4)  X3_UNIT(19, X3_LDB(22), 5)  X3_UNIT(20, X3_LDB(23), 5)
          X3_UNIT(21, X3_LDW(8) X3_LDB(24), 4)  X3_UNIT(22, X3_LDB(25), 5)  X3_UNIT(23, X3_LDB(26), 5)
          X3_UNIT(24, , 4)                      X3_UNIT(25, , 2)            X3_UNIT(26, , 0)
#undef X3_UNIT
          // the accumulators were written by inline-asm MFMAs, which the compiler's hazard recogniser does not see: the last
          // of them must have retired before VALU code (emit, the Xp <- Xn copies) reads its result (11+ wait states)
          // — and every later read is made to depend on this statement (operands tied): without that the compiler may place
          // emit's first VALU reads in front of the wait states, the asm having no visible connection to the accumulators
          asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Xp[0]), "+v"(Xp[1]), "+v"(Xp[2]), "+v"(Xn[0]), "+v"(Xn[1]), "+v"(Xn[2]), "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]) :: "memory");
          emit(p - 1);                   // X[p-1] is complete once plane p has contributed its kd=2 taps
#pragma unroll
          for (int f = 0; f < 3; ++f) Xp[f] = Xn[f];
          // ... and the copies must be done before the next plane's MFMAs overwrite Xn (VALU reads its operands at issue)
        }
        // LDS reads pinned to their side of the barrier in both directions (see conv0_sweep.hip)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // plane D is zero padding: X[D-1] only lacks the (zero) kd=2 contribution
#pragma unroll
      for (int f = 0; f < 3; ++f) Y[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      emit(D - 1);
      pg0 += D;
    }
  }
}

// fp32 weights in a 16-bit kernel's fragment order ([operands][8 values]: conv0_sweep_pack's [18][16 rows][4 k-groups][8], or
// conv3d_tile_pack in the BF16 geometry) -> device array of bf16x8 operands: all hi operands, followed by all lo operands
int conv0_sweep_x3_upload(const std::vector<float>& packed, void** dev) {
  RGBM_REQUIRE(!packed.empty() && packed.size() % 8 == 0, "split operand upload: element count must be a multiple of 8");
  const size_t nop = packed.size() / 8;
  std::vector<unsigned short> h(2 * packed.size());
  auto bf = [](float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  for (size_t i = 0; i < packed.size(); ++i) {
    const unsigned short hi = bf(packed[i]);
    const unsigned uh = (unsigned)hi << 16;
    float hf;
    memcpy(&hf, &uh, 4);
    h[i] = hi;
    h[nop * 8 + i] = bf(packed[i] - hf);
  }
  RGBM_CHECK_HIP(hipMalloc(dev, h.size() * 2));
  RGBM_CHECK_HIP(hipMemcpy(*dev, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// t.feat: the fp32 feature map; t.wgt: conv0_sweep_x3_upload's array; t.out: bx3 [N][D][H][W][8]
int launch_conv0_sweep_x3(const Conv3dTileDesc& t, hipStream_t s) {
  Sweep3Desc d;
  d.feat = reinterpret_cast<const float*>(t.feat);
  d.wgt = reinterpret_cast<const uint4*>(t.wgt);
  d.bias = t.bias; d.homog = t.homog; d.depths = t.depths;
  d.out = reinterpret_cast<bx3_t*>(t.out);
  d.N = t.N; d.D = t.Di; d.H = t.Hi; d.W = t.Wi; d.v0 = t.v0; d.V = t.V; d.B = t.B; d.relu = t.relu;
  d.nth = (d.H + X3_TH - 1) / X3_TH; d.ntw = (d.W + X3_TW - 1) / X3_TW;
  RGBM_REQUIRE(d.feat && d.wgt && d.bias && d.homog && d.depths && d.out && d.D >= 1 && d.D <= 64 && t.Cout == 8, "conv0 sweep (bf16x3) arguments");
  RGBM_REQUIRE((long long)d.H * d.W * 128ll < (1ll << 32), "conv0 sweep (bf16x3): feature map too large for 32-bit offsets");
  const long long ntiles = (long long)d.N * d.nth * d.ntw;
  RGBM_REQUIRE(ntiles > 0 && ntiles < (1ll << 31), "conv0 sweep grid out of range");
  d.n_tiles = (int)ntiles;
  d.tile_list = t.tile_list; d.tile_count = t.tile_count;
  RGBM_REQUIRE((d.tile_list == nullptr) == (d.tile_count == nullptr), "conv0 sweep: tile list and count go together");
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv0_sweep_x3_kernel), X3_LDS)) return rc;
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  // the tile order assumes a grid that is a multiple of 8 (XCD = block % 8); small launches round up and idle blocks exit
  int grid = ntiles < n_cu ? (int)((ntiles + 7) / 8 * 8) : n_cu;
  prof_begin_launch(s, t.prof_variant, t.algo_flops, t.algo_bytes);
  hipLaunchKernelGGL(conv0_sweep_x3_kernel, dim3((unsigned)grid), dim3(X3_THREADS), X3_LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
