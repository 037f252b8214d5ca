// Weight gradient of the 'same' 5x5x5 convolutions with the 1-D Winograd F(2,5) transform along W -- the
// adjoint of conv_halo_wino_k (msk_conv_wino.hip):
//   dU_xi[kd,kh][ca][cb] = sum_{n,d,h,t} V_xi[n, d+kd-2, h+kh-2, t][ca] * Y_xi[n, d, h, t][cb]
//   V_xi = sum_j BT[xi][j] x[.., 2t-2+j]  (j = 0..5),   Y_xi = AT[0][xi] dy[.., 2t] + AT[1][xi] dy[.., 2t+1]
//   dW[kd,kh,kw] = sum_xi G[xi][kw] dU_xi[kd,kh]                      (wgrad_wino_reduce_k, with the split-K sum)
// 6 multiplications per W pair instead of 10: 0.6 of the MFMA work of wgrad_lds_mfma_k.
//
// Workgroup = 6 wavefronts, one per xi, owning one kd plane and one 32x32 (ca, cb) tile: 5 accumulators (kh) per
// wave.  Per chunk (one output depth, R rows x WS columns) the x halo rows and the dy rows are staged once in LDS
// ([voxel][32 channels]: every operand is a conflict-free ds_read_b32); a wave walks the rows of one column pair
// with a 5-row sliding window of transformed x values, so each step costs 6 + 2 LDS reads, ~13 VALU ops and
// 5 MFMAs (K = the two W pairs of the column pair: lane half = pair parity).
#include "msk_conv.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int R = 8, WS = 16, P = 2;
constexpr int XR = R + 2 * P, XW = WS + 2 * P;  // 12 x 20 halo
constexpr int RPX = XW * 32 + 16, RPY = WS * 32 + 16;  // row pitches (floats): +16 -> the 4 rows of an MFMA K group
                                                       // land on 4 different 16-bank groups (conflict-free ds_read_b32)
constexpr int NT = 256;

typedef float f2 __attribute__((ext_vector_type(2)));
// 6-point input transform of two independent rows at once (packed fp32: v_pk_fma_f32 / v_pk_add_f32)
__device__ __forceinline__ void wino_bt2(const f2 x0, const f2 x1, const f2 x2, const f2 x3, const f2 x4, const f2 x5,
                                         f2 (&v)[6]) {
  const f2 c4 = {4.f, 4.f}, c5 = {5.f, 5.f}, c2 = {2.f, 2.f};
  v[0] = c4 * x0 + (x4 - c5 * x2);
  const f2 pa = x4 - c4 * x2, qa = x3 - c4 * x1;
  v[1] = pa + qa;
  v[2] = pa - qa;
  const f2 pb = x4 - x2, qb = c2 * (x3 - x1);
  v[3] = pb + qb;
  v[4] = pb - qb;
  v[5] = c4 * x1 + (x5 - c5 * x3);
}

// Workgroup = 4 wavefronts = the four 16x16 quadrants of one 32x32 (ca, cb) tile of one kd plane; a wavefront keeps
// ALL 30 (xi, kh) accumulators of its quadrant (v_mfma_f32_16x16x4_f32: 4 registers each = 120 VGPRs).  The MFMA K
// dimension (4) runs over 4 consecutive OUTPUT ROWS of one W pair: lane group g = lane >> 4 transforms the x rows
// r0+g .. r0+g+4 (6 quads -> 6 xi values each, 13 VALU ops per row) and its dy row, then issues 30 MFMAs:
// 32 LDS reads and ~70 VALU ops per 30 MFMAs (the first version with one xi per wavefront and 32x32x2 MFMAs had
// 5 MFMAs per 8 reads / 13 VALU ops and reached only 46 % MFMA utilisation with unbalanced 6-wave workgroups).
__global__ void __launch_bounds__(NT, 2)
wgrad_wino_k(WGrad g, int splits, int chunks_total, int chunks_per_split, float* __restrict__ partial) {
  __shared__ float xs[XR * RPX];
  __shared__ float dys[R * RPY];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int kd = b / ca_tiles;
  const int split = blockIdx.y;
  const int D = g.BD, H = g.BH, W = g.BW;
  const int hblocks = (H + R - 1) / R, wblocks = (W + WS - 1) / WS;
  const int qa = (wave & 1) * 16, qb = (wave >> 1) * 16;  // channel offsets of this wave's quadrant inside the tile

  f32x4 acc[6][5];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[x][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int c_begin = split * chunks_per_split;
  int c_end = c_begin + chunks_per_split;
  if (c_end > chunks_total) c_end = chunks_total;
  constexpr int XITEMS = XR * XW * 8, DITEMS = R * WS * 8;

  // Software pipeline: the global loads of chunk i+1 are issued into registers before the MFMA loop of chunk i and
  // written to LDS after it (12 float4 = 48 VGPRs; the kernel runs 2 waves per SIMD either way).
  constexpr int XPT = (XITEMS + NT - 1) / NT, DPT = (DITEMS + NT - 1) / NT;  // 8 + 4 float4 per thread
  float4 px[XPT], pd[DPT];
  auto next_valid = [&](int ch) {
    while (ch < c_end) {
      const int d = (ch / (wblocks * hblocks)) % D;
      if ((unsigned)(d + kd - P) < (unsigned)D) break;
      ++ch;
    }
    return ch;
  };
  auto load = [&](int ch) {
    int t = ch;
    const int wb = t % wblocks;
    t /= wblocks;
    const int hb = t % hblocks;
    t /= hblocks;
    const int d = t % D;
    const int n = t / D;
    const int id = d + kd - P;
    const int h0 = hb * R, w0 = wb * WS;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int it = tid + i * NT;
      const int q = it & 7, v = it >> 3;
      const int row = v / XW, col = v % XW;
      const int ih = h0 - P + row, iw = w0 - P + col;
      const int c0 = cat * 32 + q * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < XITEMS && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && c0 < g.CA)
        val = *reinterpret_cast<const float4*>(g.A + ((long)n * g.vsn + (long)id * g.vsd + (long)ih * g.vsh + (long)iw * g.vsw) * g.ald + c0);
      px[i] = val;
    }
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int it = tid + i * NT;
      const int q = it & 7, v = it >> 3;
      const int row = v / WS, col = v % WS;
      const int oh = h0 + row, ow = w0 + col;
      const int c0 = cbt * 32 + q * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < DITEMS && oh < H && ow < W && c0 < g.CB)
        val = *reinterpret_cast<const float4*>(g.B + ((long)n * g.vsn + (long)d * g.vsd + (long)oh * g.vsh + (long)ow * g.vsw) * g.bld + c0);
      pd[i] = val;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int it = tid + i * NT;
      if (it < XITEMS) {
        const int q = it & 7, v = it >> 3;
        *reinterpret_cast<float4*>(&xs[(v / XW) * RPX + (v % XW) * 32 + q * 4]) = px[i];
      }
    }
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int it = tid + i * NT;
      if (it < DITEMS) {
        const int q = it & 7, v = it >> 3;
        *reinterpret_cast<float4*>(&dys[(v / WS) * RPY + (v % WS) * 32 + q * 4]) = pd[i];
      }
    }
  };

  int ch = next_valid(c_begin);
  if (ch < c_end) load(ch);
  while (ch < c_end) {
    __syncthreads();  // every wave finished reading the previous chunk
    store();
    __syncthreads();
    const int nxt = next_valid(ch + 1);
    if (nxt < c_end) load(nxt);  // in flight during the MFMA loop below

    const float* xlane = &xs[lg * RPX + qa + li];
    const float* dlane = &dys[lg * RPY + qb + li];
    // The (row group, W pair, kh) iterations are software-pipelined by hand: the six LDS reads of the NEXT x row (and
    // the two dy reads of the next W pair) are issued before the six MFMAs of the current row, so the LDS latency and
    // the 13-op transform overlap the matrix pipe (without this every kh exposed ~170 cycles before 192 MFMA cycles:
    // PMC showed the MFMA pipe 60 % busy with almost no wave waiting on memory).
    constexpr int STEPS = (R / 4) * (WS / 2);
    float nx[6], ny0, ny1;
    {
      const float* p = xlane;
#pragma unroll
      for (int j = 0; j < 6; ++j) nx[j] = p[j * 32];
      ny0 = dlane[0];
      ny1 = dlane[32];
    }
#pragma unroll 1
    for (int st = 0; st < STEPS; ++st) {
      const int rg = st / (WS / 2), tp = st % (WS / 2);
      const int st1 = st + 1 < STEPS ? st + 1 : st;  // the last prefetch re-reads a valid address and is discarded
      const int rg1 = st1 / (WS / 2), tp1 = st1 % (WS / 2);
      const float* xp = xlane + rg * 4 * RPX + tp * 64;
      const float y0 = ny0, y1 = ny1;
      const float ys[6] = {y0, y0 + y1, y0 - y1, fmaf(2.f, y1, y0), fmaf(-2.f, y1, y0), y1};
      // this step's five x rows (the first one was prefetched), then the prefetch of the next step's first row
      float xr[5][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) xr[0][j] = nx[j];
#pragma unroll
      for (int kh = 1; kh < 5; ++kh) {
        const float* p = xp + kh * RPX;
#pragma unroll
        for (int j = 0; j < 6; ++j) xr[kh][j] = p[j * 32];
      }
      {
        const float* p = xlane + rg1 * 4 * RPX + tp1 * 64;
#pragma unroll
        for (int j = 0; j < 6; ++j) nx[j] = p[j * 32];
        const float* dp = dlane + rg1 * 4 * RPY + tp1 * 64;
        ny0 = dp[0];
        ny1 = dp[32];
      }
      // V = BT x: rows (0,1) and (2,3) as packed pairs (v_pk_* instructions), row 4 scalar: 39 instead of 65 VALU ops
      f2 va[6], vb[6];
      float vc[6];
      wino_bt2((f2){xr[0][0], xr[1][0]}, (f2){xr[0][1], xr[1][1]}, (f2){xr[0][2], xr[1][2]}, (f2){xr[0][3], xr[1][3]},
               (f2){xr[0][4], xr[1][4]}, (f2){xr[0][5], xr[1][5]}, va);
      wino_bt2((f2){xr[2][0], xr[3][0]}, (f2){xr[2][1], xr[3][1]}, (f2){xr[2][2], xr[3][2]}, (f2){xr[2][3], xr[3][3]},
               (f2){xr[2][4], xr[3][4]}, (f2){xr[2][5], xr[3][5]}, vb);
      {
        const float x0 = xr[4][0], x1 = xr[4][1], x2 = xr[4][2], x3 = xr[4][3], x4 = xr[4][4], x5 = xr[4][5];
        const float pa = fmaf(-4.f, x2, x4), qa_ = fmaf(-4.f, x1, x3);
        const float pb = x4 - x2, qb_ = 2.f * (x3 - x1);
        vc[0] = fmaf(4.f, x0, fmaf(-5.f, x2, x4));
        vc[1] = pa + qa_;
        vc[2] = pa - qa_;
        vc[3] = pb + qb_;
        vc[4] = pb - qb_;
        vc[5] = fmaf(4.f, x1, fmaf(-5.f, x3, x5));
      }
#pragma unroll
      for (int x = 0; x < 6; ++x) acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[x].x, ys[x], acc[x][0], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 6; ++x) acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[x].y, ys[x], acc[x][1], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 6; ++x) acc[x][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[x].x, ys[x], acc[x][2], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 6; ++x) acc[x][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[x].y, ys[x], acc[x][3], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 6; ++x) acc[x][4] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[x], ys[x], acc[x][4], 0, 0, 0);
    }
    ch = nxt;
  }

  // D: col = lane & 15 (cb), row = (lane >> 4) * 4 + reg (ca);  partial[split][xi][kd*5 + kh][ca][cb]
  const int cb = cbt * 32 + qb + li;
  if (cb < g.CB) {
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int oca = cat * 32 + qa + lg * 4 + j;
          if (oca < g.CA)
            partial[((((long)split * 6 + x) * 25 + kd * 5 + kh) * g.CA + oca) * g.CB + cb] = acc[x][kh][j];
        }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F(4,5) variant (W % 4 == 0): 8 xi planes, 40 (xi, kh) accumulators per wavefront (160 VGPRs), MFMA K = 4 output
// rows of one W QUAD: per step 5 x 8 x-reads + 4 dy reads, 5 x 26 + 12 transform ops and 40 MFMAs.  Executes 0.4 of
// the direct MACs (the F(2,5) kernel above: 0.6).  The x rows are transformed one at a time to keep the register
// count below 256.
// ---------------------------------------------------------------------------------------------------------
constexpr int VP = 36;  // floats per staged voxel (32 channels + 4): the four W quads of an MFMA K group are 4 voxels =
                        // 144 floats apart -> 16 banks apart -> conflict-free ds_read_b32 for the 4 x 16 lanes

// MFMA K (4) = the four W QUADS of one output row; a lane group g = lane >> 4 owns quad g.  Walking down the rows of
// the chunk, a lane keeps a 5-row sliding window of transformed x values, so a step costs ONE new x-row transform
// (8 LDS reads, 12-14 VALU ops for this wave's 4 xi), the dy transform of the row (2 x 4 reads) and 40 MFMAs.
// (The previous mapping, K = 4 consecutive rows of one quad, transformed five x rows per step: ~150 VALU
// instructions per 40 MFMAs, MFMA pipe 53 % busy; fp32 MFMA shares the SIMD issue with the VALU.)
__global__ void __launch_bounds__(NT, 2)
wgrad_wino4_k(WGrad g, int splits, int chunks_total, int chunks_per_split, float* __restrict__ partial, unsigned a_bytes,
              unsigned b_bytes) {
  __shared__ float xs[XR * XW * VP];
  __shared__ float dys[R * WS * VP];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int kd = b / ca_tiles;
  const int split = blockIdx.y;
  const int D = g.BD, H = g.BH, W = g.BW;
  const int hblocks = (H + R - 1) / R, wblocks = (W + WS - 1) / WS;
  // wave = (ca half, xi half): 16 ca rows x both 16-column cb halves for 4 of the 8 xi planes
  const int qa = (wave & 1) * 16, xh = wave >> 1;

  f32x4 acc[2][4][5];  // [cb half][xi slot][kh]
#pragma unroll
  for (int hb_ = 0; hb_ < 2; ++hb_)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[hb_][x][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int c_begin = split * chunks_per_split;
  int c_end = c_begin + chunks_per_split;
  if (c_end > chunks_total) c_end = chunks_total;
  constexpr int XITEMS = XR * XW * 8, DITEMS = R * WS * 8;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);

  for (int ch = c_begin; ch < c_end; ++ch) {
    int t = ch;
    const int wb = t % wblocks;
    t /= wblocks;
    const int hb = t % hblocks;
    t /= hblocks;
    const int d = t % D;
    const int n = t / D;
    const int id = d + kd - P;
    if ((unsigned)id >= (unsigned)D) continue;  // block-uniform
    const int h0 = hb * R, w0 = wb * WS;
    const unsigned a_plane = (unsigned)((((long)n * g.vsn + (long)id * g.vsd) * g.ald) * 4);
    const unsigned b_plane = (unsigned)((((long)n * g.vsn + (long)d * g.vsd) * g.bld) * 4);
    __syncthreads();
    for (int base = 0; base < XITEMS; base += 4 * NT) {
      float4 tmp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        const int q = it & 7, v = it >> 3;
        const int row = v / XW, col = v % XW;
        const int ih = h0 - P + row, iw = w0 - P + col;
        const int c0 = cat * 32 + q * 4;
        // wave-uniform (n, plane) part of the address in the scalar offset, the lane part in 32 bits; columns outside
        // the volume read zeros through an out-of-range offset (64-bit pointer arithmetic per element: ~65 slow
        // VALU instructions per chunk)
        const bool ok = it < XITEMS && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && c0 < g.CA;
        const unsigned voff = ok ? (unsigned)((ih * g.vsh + iw * g.vsw) * g.ald + c0) * 4u : 0xFFFFFFF0u;
        tmp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, (int)a_plane, 0));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        if (it < XITEMS) *reinterpret_cast<float4*>(&xs[(it >> 3) * VP + (it & 7) * 4]) = tmp[i];
      }
    }
    for (int base = 0; base < DITEMS; base += 4 * NT) {
      float4 tmp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        const int q = it & 7, v = it >> 3;
        const int row = v / WS, col = v % WS;
        const int oh = h0 + row, ow = w0 + col;
        const int c0 = cbt * 32 + q * 4;
        const bool ok = it < DITEMS && oh < H && ow < W && c0 < g.CB;
        const unsigned voff = ok ? (unsigned)((oh * g.vsh + ow * g.vsw) * g.bld + c0) * 4u : 0xFFFFFFF0u;
        tmp[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)voff, (int)b_plane, 0));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        if (it < DITEMS) *reinterpret_cast<float4*>(&dys[(it >> 3) * VP + (it & 7) * 4]) = tmp[i];
      }
    }
    __syncthreads();

    const float* xlane = &xs[4 * lg * VP + qa + li];   // halo columns 4g .. 4g+7 of this lane's W quad
    const float* dlane = &dys[4 * lg * VP + li];       // output columns 4g .. 4g+3
    auto xform = [&](int row, float (&v)[4]) {
      const float* p = xlane + row * XW * VP;
      const float d0 = p[0], d1 = p[VP], d2 = p[2 * VP], d3 = p[3 * VP], d4 = p[4 * VP], d5 = p[5 * VP], d6 = p[6 * VP],
                  d7 = p[7 * VP];
      if (xh == 0) {  // xi 0, 7, 1, 2
        v[0] = (d6 - d0) + 5.25f * (d2 - d4);
        v[1] = (d7 - d1) + 5.25f * (d3 - d5);
        const float t1 = fmaf(-4.25f, d4, d2 + d6), t2 = fmaf(-4.25f, d3, d1 + d5);
        v[2] = t1 + t2;
        v[3] = t1 - t2;
      } else {        // xi 3, 4, 5, 6
        const float t3 = fmaf(-1.25f, d4, fmaf(0.25f, d2, d6)), t4 = fmaf(2.f, d5, fmaf(-2.5f, d3, 0.5f * d1));
        const float t5 = fmaf(-5.f, d4, fmaf(4.f, d2, d6)), t6 = fmaf(0.5f, d5, fmaf(-2.5f, d3, 2.f * d1));
        v[0] = t3 + t4;
        v[1] = t3 - t4;
        v[2] = t5 + t6;
        v[3] = t5 - t6;
      }
    };
    // RPT output rows per trip: the window of transformed x rows (r .. r+3+RPT, this wave's 4 xi) is shifted once per
    // RPT * 40 MFMAs (16 v_mov) instead of once per 40.  RPT = 4 spills (256 VGPRs + 176 B of scratch, 1.5x slower).
    constexpr int RPT = 2;
    float win[4 + RPT][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xform(k, win[k]);
    auto row_step = [&](int r, int w0_) {
      float ys[2][4];
#pragma unroll
      for (int hb_ = 0; hb_ < 2; ++hb_) {
        const float* dp = dlane + r * WS * VP + hb_ * 16;
        const float y0 = dp[0], y1 = dp[VP], y2 = dp[2 * VP], y3 = dp[3 * VP];
        if (xh == 0) {
          const float e = y0 + y2, o = y1 + y3;
          ys[hb_][0] = y0;
          ys[hb_][1] = y3;
          ys[hb_][2] = e + o;
          ys[hb_][3] = e - o;
        } else {
          const float e2 = fmaf(4.f, y2, y0), o2 = fmaf(8.f, y3, 2.f * y1);
          const float e3 = fmaf(0.25f, y2, y0), o3 = fmaf(0.125f, y3, 0.5f * y1);
          ys[hb_][0] = e2 + o2;
          ys[hb_][1] = e2 - o2;
          ys[hb_][2] = e3 + o3;
          ys[hb_][3] = e3 - o3;
        }
      }
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          acc[0][x][kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[w0_ + kh][x], ys[0][x], acc[0][x][kh], 0, 0, 0);
          acc[1][x][kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[w0_ + kh][x], ys[1][x], acc[1][x][kh], 0, 0, 0);
        }
    };
    static_assert(R % RPT == 0, "whole trips");
#pragma unroll 1
    for (int r = 0; r < R; r += RPT) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        xform(r + 4 + j, win[4 + j]);
        row_step(r + j, j);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) win[k][x] = win[k + RPT][x];
    }
  }

  // xi plane of slot x: half 0 -> {0, 7, 1, 2}, half 1 -> {3, 4, 5, 6}
#pragma unroll
  for (int hb_ = 0; hb_ < 2; ++hb_) {
    const int cb = cbt * 32 + hb_ * 16 + li;
    if (cb < g.CB) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int xi = xh == 0 ? (x == 0 ? 0 : (x == 1 ? 7 : x - 1)) : 3 + x;
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int oca = cat * 32 + qa + lg * 4 + j;
            if (oca < g.CA)
              partial[((((long)split * 8 + xi) * 25 + kd * 5 + kh) * g.CA + oca) * g.CB + cb] = acc[hb_][x][kh][j];
          }
      }
    }
  }
}

// block = 64 consecutive (ca, cb) elements of one (kd, kh) row x 4 split slices: every slab read is a coalesced 256-byte
// row, the four slices are combined through LDS in a fixed order, slice 0 applies G^T and writes the 5 kw taps.
// (The first version gave each output element to one thread that walked all splits x 8 planes alone: 25.6 K threads
// for a 32 -> 32 layer, 0.18 ms per call, latency bound.)
__global__ void __launch_bounds__(256)
wgrad_wino4_reduce_k(const float* __restrict__ partial, int splits, int CA, int CB, int tsd, int tsh, int tsw,
                     float* __restrict__ dw, int accumulate) {
  const double G[8][5] = {{-1, 0, 0, 0, 0},
                          {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                          {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                          {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                          {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                          {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                          {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                          {0, 0, 0, 0, 1}};
  __shared__ double sh[3][64][8];
  const long plane = (long)CA * CB;
  const long per = 8L * 25 * plane;
  const long total = 25 * plane;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long idx = base + lane;
    double u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long e = 0;
    int row = 0;
    if (idx < total) {
      e = idx % plane;
      row = (int)(idx / plane);
      const float* p = partial + (long)row * plane + e;
      for (int k = slice; k < splits; k += 4) {
        const float* pk = p + (long)k * per;
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) u[xi] += pk[(long)xi * 25 * plane];
      }
    }
    if (slice > 0) {
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) sh[slice - 1][lane][xi] = u[xi];
    }
    __syncthreads();
    if (slice == 0 && idx < total) {
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) u[xi] = (u[xi] + sh[0][lane][xi]) + (sh[1][lane][xi] + sh[2][lane][xi]);
      const int cb = (int)(e % CB), ca = (int)(e / CB);
      float* o = dw + ((long)cb * CA + ca) * 125 + (row / 5) * tsd + (row % 5) * tsh;  // logical (kd, kh) -> canonical tap
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        double s_ = 0.0;
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) s_ += G[xi][kw] * u[xi];
        o[kw * tsw] = accumulate ? o[kw * tsw] + (float)s_ : (float)s_;
      }
    }
    __syncthreads();
  }
}

// Few slabs, large planes (the 128- and 256-channel layers: 1-3 splits of 13-52 MB): no slices to combine, so a thread
// takes FOUR consecutive cb (one 16-byte load per plane) and walks the slabs itself -- same summation order as the
// sliced kernel for splits <= 3.  (Sliced kernel at splits = 1: 3 of 4 wavefronts idle, 4-byte loads, 0.13 ms per
// 256 x 256 layer for 84 MB of traffic.)
__global__ void __launch_bounds__(256)
wgrad_wino4_reduce_few_k(const float* __restrict__ partial, int splits, int CA, int CB, int tsd, int tsh, int tsw,
                         float* __restrict__ dw, int accumulate) {
  const double G[8][5] = {{-1, 0, 0, 0, 0},
                          {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                          {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                          {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                          {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                          {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                          {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                          {0, 0, 0, 0, 1}};
  const long plane4 = (long)CA * CB / 4;
  const long per4 = 8L * 25 * plane4;
  const long total4 = 25 * plane4;
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
    const long e4 = idx % plane4;
    const int row = (int)(idx / plane4);
    double u[8][4];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      const float4 v = p4[((long)xi * 25 + row) * plane4 + e4];
      u[xi][0] = v.x; u[xi][1] = v.y; u[xi][2] = v.z; u[xi][3] = v.w;
    }
    for (int k = 1; k < splits; ++k) {
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) {
        const float4 v = p4[(long)k * per4 + ((long)xi * 25 + row) * plane4 + e4];
        u[xi][0] += v.x; u[xi][1] += v.y; u[xi][2] += v.z; u[xi][3] += v.w;
      }
    }
    const long e = e4 * 4;
    const int cb = (int)(e % CB), ca = (int)(e / CB);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* o = dw + ((long)(cb + j) * CA + ca) * 125 + (row / 5) * tsd + (row % 5) * tsh;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        double s_ = 0.0;
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) s_ += G[xi][kw] * u[xi][j];
        o[kw * tsw] = accumulate ? o[kw * tsw] + (float)s_ : (float)s_;
      }
    }
  }
}

// dw[cb][ca][(kd,kh,kw)] (+)= sum_xi G[xi][kw] * sum_split P[split][xi][kd*5+kh][ca][cb]   (fixed order, double)
__global__ void __launch_bounds__(256)
wgrad_wino_reduce_k(const float* __restrict__ partial, int splits, int CA, int CB, int tsd, int tsh, int tsw,
                    float* __restrict__ dw, int accumulate) {
  const double G[6][5] = {{0.25, 0, 0, 0, 0},
                          {-1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6},
                          {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6, 1.0 / 3, 2.0 / 3},
                          {1.0 / 24, -1.0 / 12, 1.0 / 6, -1.0 / 3, 2.0 / 3},
                          {0, 0, 0, 0, 1}};
  const long plane = (long)CA * CB;         // one (xi, row) plane
  const long per = 6L * 25 * plane;         // one split
  const long total = 25 * plane;            // outputs before the kw expansion
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long e = idx % plane;             // ca*CB + cb
    const int row = (int)(idx / plane);     // kd*5 + kh
    double u[6];
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
      const float* p = partial + ((long)xi * 25 + row) * plane + e;
      double s0 = 0.0, s1 = 0.0;
      int k = 0;
      for (; k + 1 < splits; k += 2) {
        s0 += p[(long)k * per];
        s1 += p[(long)(k + 1) * per];
      }
      if (k < splits) s0 += p[(long)k * per];
      u[xi] = s0 + s1;
    }
    const int cb = (int)(e % CB), ca = (int)(e / CB);
    float* o = dw + ((long)cb * CA + ca) * 125 + (row / 5) * tsd + (row % 5) * tsh;
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      double s = 0.0;
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) s += G[xi][kw] * u[xi];
      o[kw * tsw] = accumulate ? o[kw * tsw] + (float)s : (float)s;
    }
  }
}

}  // namespace

int msk_wgrad_wino(msk_ctx* ctx, const WGrad& g_in) {
  if (!(g_in.kd == 5 && g_in.kh == 5 && g_in.kw == 5 && g_in.sd == 1 && g_in.sh == 1 && g_in.sw == 1 && g_in.pd == 2 &&
        g_in.ph == 2 && g_in.pw == 2))
    return 0;
  if (!(g_in.AD == g_in.BD && g_in.AH == g_in.BH && g_in.AW == g_in.BW)) return 0;
  if (!(g_in.ald % 4 == 0 && g_in.bld % 4 == 0 && g_in.CA % 4 == 0 && g_in.CB % 4 == 0 && ((uintptr_t)g_in.A) % 16 == 0 &&
        ((uintptr_t)g_in.B) % 16 == 0))
    return 0;
  // Logical axes (d: planes / kd tasks, h: 8-row chunks, w: 16-column chunks + transform) = the permutation of the
  // tensor's axes with the least chunk padding among those whose w is even and >= 16 (narrower volumes keep the direct
  // LDS kernel); F(4,5) (w % 4 == 0) first, the identity on ties.  MRI slabs (W = 12 / 9 / 8 / 4 / 2) transform along H.
  const int dims[3] = {g_in.BD, g_in.BH, g_in.BW};
  static const int kPerms[6][3] = {{0, 1, 2}, {1, 0, 2}, {0, 2, 1}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};
  int best = -1, best_f45 = 0;
  long best_cost = 0;
  for (int i = 0; i < 6; ++i) {
    const int ld_ = dims[kPerms[i][0]], lh_ = dims[kPerms[i][1]], lw_ = dims[kPerms[i][2]];
    if (lw_ % 2 || lw_ < 8) continue;   // w = 8: half of a 16-column chunk is padding, still ahead of the direct kernel
    const int f = (lw_ % 4 == 0) && ctx->conv_impl != 14;  // 14 = F(2,5) only (A/B)
    const long cost = (long)ld_ * (((lh_ + R - 1) / R) * R) * (((lw_ + WS - 1) / WS) * WS);
    if (best < 0 || f > best_f45 || (f == best_f45 && cost < best_cost)) {
      best = i;
      best_f45 = f;
      best_cost = cost;
    }
  }
  if (best < 0) return 0;
  const int* pm = kPerms[best];
  const int vstr[3] = {g_in.BH * g_in.BW, g_in.BW, 1}, tstr[3] = {25, 5, 1};
  WGrad g = g_in;
  g.BD = g.AD = dims[pm[0]]; g.BH = g.AH = dims[pm[1]]; g.BW = g.AW = dims[pm[2]];
  g.vsn = (long)g_in.BD * g_in.BH * g_in.BW; g.vsd = vstr[pm[0]]; g.vsh = vstr[pm[1]]; g.vsw = vstr[pm[2]];
  const int tsd = tstr[pm[0]], tsh = tstr[pm[1]], tsw = tstr[pm[2]];
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const long chunks = (long)g.N * g.BD * ((g.BH + R - 1) / R) * ((g.BW + WS - 1) / WS);
  const long tasks = 5L * ca_tiles * cb_tiles;
  const bool f45 = best_f45 != 0;
  const int nxi = f45 ? 8 : 6;
  const size_t per = (size_t)nxi * 25 * g.CA * g.CB * sizeof(float);
  long splits;
  if (ctx->wgrad_wino_rounds > 0) {  // manual: ~rounds/2 waves of the 2 resident workgroups per CU
    splits = ((long)ctx->num_cu * ctx->wgrad_wino_rounds + tasks - 1) / tasks;
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    while (splits > 1 && splits * per > ((size_t)1 << 30)) --splits;
  } else {
    // All workgroups do the same work, two are resident per CU (242 VGPRs), so the kernel advances in waves of
    // slots = 2 * num_cu workgroups: time ~ ceil(tasks * splits / slots) * chunks_per_split * t_chunk, and every
    // split adds one slab to write and to reduce.  Measured on MI355X: t_chunk ~13.5 us with both slots busy, the
    // reduce ~1 ns per KB of slab.  (A fixed 3 waves -- the first policy -- cost 4-40 % per layer: 515 workgroups
    // on 512 slots is two waves.)
    const long slots = 2L * ctx->num_cu;
    long smax = chunks;
    if (smax > (long)(((size_t)1 << 30) / per)) smax = (long)(((size_t)1 << 30) / per);
    if (smax > slots * 6 / tasks + 1) smax = slots * 6 / tasks + 1;
    if (smax < 1) smax = 1;
    double best = 1e30;
    splits = 1;
    for (long s_ = 1; s_ <= smax; ++s_) {
      const long c_ = (chunks + s_ - 1) / s_;
      if ((chunks + c_ - 1) / c_ != s_) continue;
      const long waves = (tasks * s_ + slots - 1) / slots;
      const double cost = (double)waves * c_ * 13.5e-6 + (double)s_ * per * 1e-12;
      if (cost < best) {
        best = cost;
        splits = s_;
      }
    }
  }
  const int cps = (int)((chunks + splits - 1) / splits);
  splits = (chunks + cps - 1) / cps;
  if (chunks > 0x7fffffff) return 0;
  const size_t abytes = (size_t)g.N * g.BD * g.BH * g.BW * g.ald * sizeof(float);
  const size_t bbytes = (size_t)g.N * g.BD * g.BH * g.BW * g.bld * sizeof(float);
  if (abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;  // 32-bit buffer offsets (run_wgrad chunks the batch)
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  {
    const char* tag = f45 ? "wgrad_wino4" : "wgrad_wino";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "%s[ca=%d,cb=%d,M=%ld,splits=%ld]", tag, g.CA, g.CB, (long)g.N * g.BD * g.BH * g.BW, splits);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    if (f45)
      hipLaunchKernelGGL(wgrad_wino4_k, dim3((unsigned)tasks, (unsigned)splits), dim3(NT), 0, ctx->stream, g, (int)splits,
                         (int)chunks, cps, partial, (unsigned)abytes, (unsigned)bbytes);
    else
      hipLaunchKernelGGL(wgrad_wino_k, dim3((unsigned)tasks, (unsigned)splits), dim3(NT), 0, ctx->stream, g, (int)splits,
                         (int)chunks, cps, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    const char* rtag = "wgrad_wino_reduce";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "wgrad_wino_reduce[ca=%d,cb=%d,splits=%ld,f45=%d]", g.CA, g.CB, splits, (int)f45);
      rtag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, rtag);
    const long total = 25L * g.CA * g.CB;
    long blocks = f45 ? (total + 63) / 64 : (total + 255) / 256;
    if (blocks > 32L * ctx->num_cu) blocks = 32L * ctx->num_cu;
    if (f45 && splits <= 3 && ((uintptr_t)partial) % 16 == 0) {
      long fb = (total / 4 + 255) / 256;
      if (fb > 32L * ctx->num_cu) fb = 32L * ctx->num_cu;
      hipLaunchKernelGGL(wgrad_wino4_reduce_few_k, dim3((unsigned)fb), dim3(256), 0, ctx->stream, (const float*)partial,
                         (int)splits, g.CA, g.CB, tsd, tsh, tsw, g.dw, g.accumulate);
    } else if (f45)
      hipLaunchKernelGGL(wgrad_wino4_reduce_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)partial,
                         (int)splits, g.CA, g.CB, tsd, tsh, tsw, g.dw, g.accumulate);
    else
      hipLaunchKernelGGL(wgrad_wino_reduce_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)partial,
                         (int)splits, g.CA, g.CB, tsd, tsh, tsw, g.dw, g.accumulate);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 1;
}
