// Attention backward, round 4: ONE WORKGROUP = ONE (sequence, head), everything that the four waves of the v3 grouped kernel
// computed four times is computed once and shared through LDS (included by se_attn.hip; scaled split-fp16 arithmetic only).
//
// What the v3 kernel (attn_bwd3_kernel<6, true, F16, RING>) spent per query tile and wave (ISA census, tools/isa_blocks.py): 1 060
// vector instructions for 153 MFMAs -- 324 of them address arithmetic and splits of the SAME query-side operands in all four
// waves, 7 offset tiles per wave (28 per workgroup) for the 22 the query tile touches, masks for the strip cells of other waves'
// keys, and 242 VGPRs = two waves per SIMD.  Here:
//   * the query tile's Q and dO rows are loaded and split ONCE (the loader role rotates over the waves; raw loads are requested a
//     whole tile ahead and wait in registers of the loader only during its own key phase), staged as fp16 (hi, lo) images from
//     which every wave takes its row fragments (ds_read_b64) and column fragments (ds_read_b64_tr_b16), plus lse / delta rows;
//   * ONE offset strip per workgroup: U[q][delta] for the nkt + 1 offset tiles ("positions" j = 0 .. nkt, tile D = qt - j) of the
//     query tile; a key step reads its skewed cells and overwrites them with dS in place as before, but the strip is consumed
//     (dQ += W E, dE += W^T Q) once per position, not once per wave, and only positions 0 and nkt need a key mask;
//   * offset tiles are OWNED: tile D belongs to class D mod M, M = ceil((nkt + 1) / 2), a class to one wave for the whole kernel.
//     A class has exactly two tiles in the window (positions j0 and j0 + M), so their dE accumulators stay in registers of the
//     owner from the first to the last contribution (nkt + 1 query tiles) and leave ONCE: no LDS ring, no scratch tiles summed
//     across waves, 2 nkt tile stores per (sequence, head).  The number of classes per wave is chosen on the host so that
//     key tiles + 0.9 classes is level over the waves (n = 321: 5 + 3, 5 + 3, 5 + 3, 6 + 2);
//   * K AND V are staged pre-split once per workgroup (row fragments b64, K's column fragments tr_b16): a key step issues no
//     global load and no operand split of its own inputs;
//   * LDS: K image (nkt KB) + strip (16 x (16 (nkt + 1) + 4) floats) + stage (2.2 KB) + 1 KB per wave (dS^T / W^T transposition,
//     at the end of a query tile the wave's dQ partial) = 50.6 KB at n = 321: three workgroups per CU at <= 168 VGPRs.
// Three barriers per query tile: strip written | key steps done | strip consumed + dQ partials written.
// dE tiles go to the per-item table [2 nkt][256] (slot D + nkt): attn_de_reduce_items_kernel sums them.
#pragma once

struct AttnBwd4Plan {
  int kt0[8], cnt[8];            // key tiles of wave w: kt0 .. kt0 + cnt - 1
  int cls0[8], ncls[8];          // offset-tile classes of wave w: cls0 .. cls0 + ncls - 1
  int M;                         // number of classes = ceil((nkt + 1) / 2)
};

// key tiles: the LAST nkt % nw waves take one more (the tail tile of a sequence is mostly padding: n = 321 -> one key);
// classes: levelled on their own -- the two phases of a query tile are separated by barriers, so each phase is as long as its
// most loaded wave: ceil(M / nw) classes per wave, the waves with more key tiles first in line for one class fewer
static inline AttnBwd4Plan attn_bwd4_plan(int nkt, int nw, int ncw) {
  AttnBwd4Plan p;
  const int b = nkt / nw, r = nkt % nw;
  for (int w = 0, k = 0; w < 8; ++w) {
    p.kt0[w] = k;
    p.cnt[w] = w < nw ? b + (w >= nw - r ? 1 : 0) : 0;
    k += p.cnt[w];
    p.ncls[w] = 0;
  }
  p.M = (nkt + 2) / 2;
  if ((p.M + nw - 1) / nw > ncw) { p.M = -1; return p; }    // does not fit this instantiation
  for (int w = 0, left = p.M; w < nw; ++w) {                 // ceil of what is left over the waves that are left
    p.ncls[w] = (left + (nw - w) - 1) / (nw - w);
    left -= p.ncls[w];
  }
  for (int w = 0, c0 = 0; w < 8; ++w) { p.cls0[w] = c0; c0 += p.ncls[w]; }
  return p;
}
template <int NW, int NKTM> static inline size_t attn_bwd4_lds(int nkt) {        // K image | two strips | stage | Dimg | dQ slots
  return (size_t)nkt * 1024 + (size_t)2 * 16 * (16 * (NKTM + 1) + 4) * 4 + 2048 + 128 + 2 * NW * 1024;
}
template <int V> struct IC4_ { static constexpr int value = V; };
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t bwd4_rsrc_(const void* p) {       // wave-uniform base, no useful bound
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((size_t)hi << 32) | lo), 0, 0x7ffffff0, 0x00020000);
}
// (hi, lo) fp16 split of four values that are already at their scale; lo = x * 1 - hi with an OPAQUE 1.0 so that the subtraction
// is ONE v_fma_mix_f32 per value (the plain form compiles to v_cvt_f32_f16 + v_sub_f32: 12 instead of 8 instructions per split)
static __device__ __forceinline__ S3 split2h1(float x0, float x1, float x2, float x3, float one) {
  const unsigned h0 = pk_f16a(x0, x1), h1 = pk_f16a(x2, x3);
  const f16x2a a = __builtin_bit_cast(f16x2a, h0), b = __builtin_bit_cast(f16x2a, h1);
  x0 = __builtin_fmaf(x0, one, -(float)a[0]); x1 = __builtin_fmaf(x1, one, -(float)a[1]);
  x2 = __builtin_fmaf(x2, one, -(float)b[0]); x3 = __builtin_fmaf(x3, one, -(float)b[1]);
  const unsigned l0 = pk_f16a(x0, x1), l1 = pk_f16a(x2, x3);
  S3 s;
  s.v = (u32x6){0u, 0u, h0, h1, l0, l1};
  return s;
}

// KPW: key tiles per wave the register arrays are built for; NCW: offset-tile classes per wave; NKTM: largest nkt (compile-time
// strip row stride and column origin: every LDS access of a key step is one per-wave base register + an immediate offset).
// The body is instantiated per (CN key tiles, NC classes) of a wave with XT = true -- the wave has exactly CN tiles and NC
// classes and every class has two tiles in the window at every query tile (2 M == nkt + 1): straight-line phases without a
// branch -- and once with XT = false (CN = KPW, NC = NCW: wave-uniform tests everywhere) for every other split.  The variants are
// chosen at the TOP of the kernel: a choice per phase merges the register assignments of the variants at every phase end (164
// v_mov per query tile in the first build).
//
// Two phases and two barriers per query tile qt (the strip is double-buffered):
//   P1  key steps of qt on strip[qt & 1]  (S, dP, softmax, dS -> cells; dV, dK, dQ partial) -- no global load feeds it: K comes from
//       the LDS image, V fragments stay in registers for the whole kernel, the cells from the strip.  What P2 needs from memory is
//       REQUESTED here and lands during the key steps: E column fragments of the tiles of qt and (loader wave) the raw Q / dO /
//       lse / delta rows of qt + 1, which it splits into the stage at the end of P1;
//   P2  E row fragments of the tiles of qt + 1 requested; the strip of qt consumed (dQ, dE); U of qt + 1 into strip[(qt + 1) & 1];
//       this wave's dQ partial to its slot; the fragments of qt + 1 out of the stage;
//   then the wave whose turn it is sums the dQ slots and stores the rows of qt (during everybody's next P1).
template <int NW, int KPW, int NCW, int NKTM, int CN, int NC, bool XT, bool TL>
static __device__ __forceinline__ void attn_bwd4_body(const AttnBwd4Args& a, const AttnBwd4Plan& pl, unsigned char* smem4,
                                                      const int wave, const int lane) {
  constexpr int SW = 16 * (NKTM + 1) + 4;                                // strip row stride (floats)
  const int c = lane & 15, g = lane >> 4;
  const int n = a.g.n, nkt = (n + 15) >> 4, nqt = nkt;
  unsigned char* Kimg = smem4;                                           // [nkt tiles][2 planes][16 keys][16 d] fp16 (K 2^sq)
  float* strip = reinterpret_cast<float*>(smem4 + nkt * 1024);           // [2][16 q][SW]: position j at columns 16 (NKTM - j) ..
  unsigned char* Qimg = smem4 + nkt * 1024 + 2 * 64 * SW;                // [2 planes][16 q][16 d] fp16 (Q 2^sq)
  unsigned char* Oimg = Qimg + 1024;                                     // the same for dO 2^sdo
  float* rowc = reinterpret_cast<float*>(Oimg + 1024);                   // [0..15] lse, [16..31] delta of the 16 query rows
  unsigned char* Dimg = reinterpret_cast<unsigned char*>(rowc + 32) + wave * 1024;      // wave-private: dS^T / W^T transposition
  float* dqs = rowc + 32 + NW * 256;                                     // [NW waves][256]: dQ partials of one query tile
  const int kt0 = pl.kt0[wave], cnt = XT ? CN : pl.cnt[wave], cls0 = pl.cls0[wave], ncls = XT ? NC : pl.ncls[wave], M = pl.M;

  const long item = xcd_item((int)blockIdx.x, (int)gridDim.x);
  const int head = (int)(item & 3), seq = (int)(item >> 2);
  const long base = seq_base(a.g, seq);
  const int ps = (int)a.g.pos_stride;
  const float* qb = a.QKV + base * 192 + head * 16;             // + pos * ps * 192 (+64: K, +128: V)
  const float* dob = a.dO + base * 64 + head * 16;
  const float* lseb = a.LSE + base * 4 + head;
  const float* dlb = a.Dl + base * 4 + head;
  float* dqb = a.dQKV + base * 192 + head * 16;
  float* dEs = a.dEs + item * (long)(2 * nkt) * 256;
  const int trrow = c >> 2, trcol = c & 3;
  const unsigned char* Esb = reinterpret_cast<const unsigned char*>(a.Es);
  const unsigned char* Etb = reinterpret_cast<const unsigned char*>(a.Ets);
  const long esp = (long)a.R * 32, etp = (long)a.ET * 32;
  const float l2e = 1.4426950408889634f;
  float one;
  asm volatile("s_mov_b32 %0, 1.0" : "=s"(one));

  // ---- scales (as attn_bwd3_body, F16) ----
  f16_clamp_mode_a();
  const float aq = *a.qkv_amax, ado = *a.do_amax;
  const int sq = f16_sexp_a(aq), sdo = f16_sexp_a(ado), se = f16_sexp_a(*a.e_amax);
  const int sds = f16_sexp_a(32.f * a.scale * ado * aq);
  const float sqf = exp2ia(sq), sdof = exp2ia(sdo);
  const float kU = exp2ia(sq - se);                          // strip cells U = q.E at the scale of S = q.k
  const float sc2 = a.scale * l2e * exp2ia(-2 * sq);
  const float kD = a.scale * exp2ia(sds - 13 - sdo - sq);    // (dP accumulator) -> scale (dP) 2^(sds - 13): times P 2^13 = dS 2^sds
  const float kdl = a.scale * exp2ia(sds - 13);
  const float cq1 = exp2ia(-sq - sds), cq2 = exp2ia(-se - sds), cdv = exp2ia(-sdo - 13);

  // ---- per-lane bases: a key step adds compile-time offsets only ----
  // every [16 rows][16] fp16 image (K tiles, Q / dO stage, the transposition image) is bank-swizzled: the 8-byte chunk p of row a
  // lives in slot (p + (a >> 2)) & 3 of its 32-byte row.  Rows a, a + 4, a + 8, a + 12 share their banks (8 dwords per row), so
  // the plain layout made every 16-lane ds_write_b64 group 4-way and every ds_read_b64 2-way conflicted: SQ_LDS_BANK_CONFLICT
  // was 60 % of SQ_LDS_IDX_ACTIVE and the LDS 71 % busy (profiles/r04a).  A transposed read supplies the address of ITS chunk
  // (rows 4g .. 4g + 3: slot (p + g) & 3), so the data it returns is unchanged.
  const int rfo = c * 32 + ((g + (c >> 2)) & 3) * 8;                                         // row fragment: row c, chunk g
  const int tfo = (4 * g + trrow) * 32 + ((trcol + g) & 3) * 8;                              // transposed read: row 4g + trrow, chunk trcol
  unsigned char* const Krow0 = Kimg + kt0 * 1024 + rfo;                                      // row fragment of tile kt0 (+ s * 1024)
  unsigned char* const Kcol0 = Kimg + kt0 * 1024 + tfo;                                      // column fragment (tr read)
  // cell (row 4g + r, key lane c) of key tile kt0 + s: cellL + r (SW + 1) + 16 (KPW - 1 - s)   (+ 16 SW for the odd strip)
  float* const cellL = strip + (4 * g) * SW + 16 * (NKTM - kt0 - (KPW - 1)) + 4 * g - c;
  unsigned char* const Dst = Dimg + rfo;                                                     // [a][dl] image: this lane's 4 values
  unsigned char* const Dtr = Dimg + tfo;
  // strip cells of keys outside the sequence (position 0: key = a - dl < 0; position nkt: key >= 16 nkt) still hold U: lane masks
  // of W[a = c][dl = 4g + i]
  bool z0[4], zn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { z0[i] = 4 * g + i > c; zn[i] = 4 * g + i <= c; }

  // ---- query-side stage: raw loads of one query tile (loader wave only), later split into the two images ----
  float4 lq = make_float4(0.f, 0.f, 0.f, 0.f), ldo = lq;
  float lr0 = 0.f, lr1 = 0.f;
  auto stage_load = [&](int qt_) {
    int qc = qt_ * 16 + c;
    if (qc > n - 1) qc = n - 1;
    lq = *reinterpret_cast<const float4*>(qb + (unsigned)(qc * ps * 192 + 4 * g));
    ldo = *reinterpret_cast<const float4*>(dob + (unsigned)(qc * ps * 64 + 4 * g));
    lr0 = lseb[(unsigned)(qc * ps * 4)];
    lr1 = dlb[(unsigned)(qc * ps * 4)];      // delta = rowsum(dO . O): the table the to_out input-gradient GEMM's epilogue (SE_EPI_DELTA) or attn_delta_kernel wrote
  };
  auto stage_store = [&]() {
    st_planes<true>(Qimg + rfo, 512, splitx<true>(lq, sqf));
    st_planes<true>(Oimg + rfo, 512, splitx<true>(ldo, sdof));
    if (g == 0) rowc[c] = lr0;
    if (g == 1) rowc[16 + c] = lr1;
  };
  if (wave == 0) stage_load(0);

  // ---- this wave's keys: K pre-split into the shared image, V fragments (B operand of dP = dO V^T) into registers ----
  S3 vfr[CN];
  {
    float4 k4s[CN], v4s[CN];
#pragma unroll
    for (int s = 0; s < CN; ++s) {
      int key = (kt0 + (s < cnt ? s : 0)) * 16 + c;
      if (key > n - 1) key = n - 1;
      k4s[s] = *reinterpret_cast<const float4*>(qb + (unsigned)(key * ps * 192 + 64 + 4 * g));
      v4s[s] = *reinterpret_cast<const float4*>(qb + (unsigned)(key * ps * 192 + 128 + 4 * g));
    }
#pragma unroll
    for (int s = 0; s < CN; ++s) {
      const bool kok = s < cnt && (kt0 + s) * 16 + c < n;
      const float4 k4 = make_float4(kok ? k4s[s].x : 0.f, kok ? k4s[s].y : 0.f, kok ? k4s[s].z : 0.f, kok ? k4s[s].w : 0.f);
      const float4 v4 = make_float4(kok ? v4s[s].x : 0.f, kok ? v4s[s].y : 0.f, kok ? v4s[s].z : 0.f, kok ? v4s[s].w : 0.f);
      if (s < cnt) st_planes<true>(Krow0 + s * 1024, 512, splitx<true>(k4, sqf));
      vfr[s] = splitx<true>(v4, sqf);
    }
  }
  if (wave == 0) stage_store();

  // short sequences (few key steps per query tile: the raw rows of the next tile do not arrive within one key phase): the loader
  // requests its rows TWO query tiles ahead and holds them in registers for one tile
  constexpr bool PF2 = KPW <= 4;
  if (PF2 && wave == 1 % NW && nqt > 1) stage_load(1);
  constexpr int NCA = NC > 0 ? NC : 1;                       // (array extents; NC = 0: a wave without offset tiles)
  f32x4 dk[CN], dv[CN], de0[NCA], de1[NCA];
  int j0[NCA];                                               // position of the younger tile of class ci at this query tile
#pragma unroll
  for (int s = 0; s < CN; ++s) { dk[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int ci = 0; ci < NCA; ++ci) {
    de0[ci] = (f32x4){0.f, 0.f, 0.f, 0.f}; de1[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cl = cls0 + ci;
    j0[ci] = cl == 0 ? 0 : M - cl;                           // (0 - cl) mod M
  }
  float omax = 0.f;
  // one finished offset tile: v[r] = dE[delta = 16 D + c][d = 4g + r] -> slot D + nkt of the item's table (ONE 16-byte store per lane; the
  // tables of all items are summed by attn_de_reduce_items_kernel on the leaf stream).  Round 6 measured the alternative the round-5
  // review asked for -- fp32 atomics from these registers into 32 replicas of the [R][16] table, no per-item tables, no 271 MB
  // reduction: the atomics cost the kernel +40 us per launch at n = 321 (two forms: tiles turned row-major through LDS for 256
  // contiguous bytes per instruction, and dE = W^T Q accumulated with d on the lane for four whole 64-byte rows per instruction),
  // more than the reduction (63 us, hidden on the leaf stream) saves: step +0.25 ms same-box -- profiles/r06_attn_de_ab.txt.
  auto flush = [&](const f32x4& v, int D) {
    *reinterpret_cast<float4*>(dEs + (D + nkt) * 256 + c * 16 + 4 * g) = make_float4(v[0] * cq1, v[1] * cq1, v[2] * cq1, v[3] * cq1);
  };
  // (diagnostic build switch dbg & 32: shader-clock stamps of the phases of the first 64 workgroups, behind the item tables)
  unsigned* const stamps = reinterpret_cast<unsigned*>(a.dEs + (long)gridDim.x * (2 * nkt) * 256);
  // The stamp area lies BEYOND what se_attn_bwd_workspace_bytes reserves (only tools/attn_bwd_stamps.py over-allocates for it), so
  // the stores exist only in a diagnostic build (-DSE_ATTN_STAMPS): SE_ATTN_DBG=32 in a product build writes nothing.
  auto stamp = [&](int qt_, int k) {
#ifdef SE_ATTN_STAMPS
    if ((a.dbg & 32) && blockIdx.x < 64 && lane == 0)
      stamps[((blockIdx.x * NW + wave) * nqt + qt_) * 8 + k] = (unsigned)__builtin_amdgcn_s_memtime();
#else
    (void)qt_; (void)k; (void)stamps;
#endif
  };
  // is the tile at position j of class slot ci one of this wave's (XT: always)
  auto own = [&](int ci, int j) { return XT || (ci < ncls && j <= nkt); };
  // E fragments of the offset tile at position j of query tile qt_ (clamped to a valid tile where the wave has none: the loads
  // are unconditional, the products are not)
  // (both tables: a wave-uniform tile base + ONE per-lane byte offset -- row fragments [16 D + c][4g ..], tile-major column
  // fragments [tile][d = c][4g ..] --, so that a load costs scalar arithmetic only)
  const unsigned eoff = (unsigned)(c * 16 + 4 * g) * 2u;
  auto e_rows = [&](int qt_, int j, bool ok) {
    if (!ok) j = nkt;
    const unsigned char* tb = Esb + (long)(16 * (qt_ - j) + a.maxpos) * 32;        // in range by the launch conditions
    return ld_planes<true>(tb + eoff, esp);
  };
  auto e_cols = [&](int qt_, int j, bool ok) {
    if (!ok) j = nkt;
    const unsigned char* tb = Etb + (long)(16 * (qt_ - j) + a.maxpos) * 32;
    return ld_planes<true>(tb + eoff, etp);
  };
  // U[q][delta] of one owned tile into the strip `sp` of its query tile: C[q = 4g + r][delta_local = c]
  auto u_tile = [&](float* sp, const S3& qr, const S3& es, int j) {
    const f32x4 uu = prodx<true>(qr, es, (f32x4){0.f, 0.f, 0.f, 0.f});
    float* up = sp + (4 * g) * SW + 16 * (NKTM - j) + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) up[r * SW] = uu[r] * kU;
  };
  S3 qrow, dorow, qcol, docol;
  float nlse2[4], dl4s[4];
  auto read_frags = [&](int qt_, bool with_qrow) {
    if (with_qrow) qrow = ld_planes<true>(Qimg + rfo, 512);                                  // Q[q = c][4g..]
    dorow = ld_planes<true>(Oimg + rfo, 512);
    qcol = tr_planes<true>(Qimg + tfo, 512);                                                // Q[q = 4g + i][d = c]
    docol = tr_planes<true>(Oimg + tfo, 512);
    const float4 lse4 = *reinterpret_cast<const float4*>(rowc + 4 * g);
    const float4 dl4 = *reinterpret_cast<const float4*>(rowc + 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // C-layout rows are queries 4g + r: p 2^13 = exp2((s + u) scale log2e + 13 - lse log2e); rows beyond n get -inf -> p = 0
      nlse2[j] = (qt_ * 16 + 4 * g + j < n) ? 13.f - f4c(lse4, j) * l2e : -__builtin_inff();
      dl4s[j] = f4c(dl4, j) * kdl;
    }
  };
  __syncthreads();
  // ---- prologue: fragments and strip of query tile 0 ----
  read_frags(0, true);
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = j0[ci] + M * t;
      const bool ok = own(ci, j);
      const S3 es = e_rows(0, j, ok);
      if (ok) u_tile(strip, qrow, es, j);
    }
  }
  __syncthreads();

  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 16;
    float* const sp = strip + (qt & 1) * 16 * SW;                        // this query tile's strip
    float* const spn = strip + ((qt + 1) & 1) * 16 * SW;                 // the next one's
    const bool more = qt + 1 < nqt;
    stamp(qt, 0);
    // ================= P1 =================
    // requested now, used in P2: E column fragments of this tile's positions
    int jn[NCA];
    S3 ecs[2 * NCA];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      jn[ci] = j0[ci] + 1 == M ? 0 : j0[ci] + 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) ecs[2 * ci + t] = e_cols(qt, j0[ci] + M * t, own(ci, j0[ci] + M * t));
    }
    const bool loader = more && wave == (qt + 1) % NW;                   // wave-uniform: stores the stage of tile qt + 1 at the end of P1
    if (PF2) { if (qt + 2 < nqt && wave == (qt + 2) % NW) stage_load(qt + 2); }
    else if (loader) stage_load(qt + 1);

    f32x4 dq = {0.f, 0.f, 0.f, 0.f};                                     // dQ^T[d = 4g + r][q = c]: K^T dS^T part
    f32x4 dq2 = {0.f, 0.f, 0.f, 0.f}, dq3 = {0.f, 0.f, 0.f, 0.f};        // E^T W^T part (its own scale; two chains)
    float* const cellP = cellL + (qt & 1) * 16 * SW;
    if (a.dbg & 2) {
    } else if constexpr (XT && CN > 0) {
      // CN steps: straight-line code, software-pipelined BY HAND in source order -- the compiler keeps LDS accesses that may alias
      // in program order and otherwise places every ds_read right in front of its first use (measured in the ISA of the first
      // build: eight exposed LDS round trips per key step).  So: the K row fragment and the four strip cells of step s + 1 are
      // read BEFORE the cell writes of step s (distinct cells: a (q, delta) cell belongs to exactly one key), K's column fragment
      // at the top of its step, and the dQ product of step s (operands: two transposed reads behind a write) is issued after the
      // S / dP products of step s + 1.
      S3 krow = ld_planes<true>(Krow0, 512);
      float cu[4];
      {
        const float* cell0 = cellP + 16 * (KPW - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) cu[r] = cell0[r * (SW + 1)];
      }
      S3 kcol_p = krow, dst_p = krow;                                    // (overwritten before use)
#pragma unroll
      for (int s = 0; s < CN; ++s) {
        const int kt = kt0 + s;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        prodx2<true>(qrow, krow, s4, dorow, vfr[s], dp);                 // S[q = 4g + r][key = c], dP[q][key]
        if (s > 0) dq = prodx<true>(kcol_p, dst_p, dq);                  // dQ^T += K^T dS^T of the previous step
        float* cell = cellP + 16 * (KPW - 1 - s);
        float cn_[4] = {0.f, 0.f, 0.f, 0.f};
        if (s + 1 < CN) {
          krow = ld_planes<true>(Krow0 + (s + 1) * 1024, 512);
#pragma unroll
          for (int r = 0; r < 4; ++r) cn_[r] = (cell - 16)[r * (SW + 1)];
        }
        kcol_p = tr_planes<true>(Kcol0 + s * 1024, 512);
        f32x4 pp, ds;
        const bool kv = kt * 16 + c < n;                                 // (TL: the last step of the last wave is the tail tile)
#if SE_ATTN_TWIN == 1
        (void)kv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pp[r] = s4[r]; ds[r] = dp[r]; cell[r * (SW + 1)] = cu[r]; }
        const S3 pps = twin_raw_(pp[0], pp[1], pp[2], pp[3]), dss = twin_raw_(ds[0], ds[1], ds[2], ds[3]);
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s4[r] + cu[r], sc2, nlse2[r]));                   // the cell of offset q - key
          if (TL && s == CN - 1) p = kv ? p : 0.f;                       // only the tail tile has keys >= n
          pp[r] = p;
          ds[r] = p * fmaf(dp[r], kD, -dl4s[r]);
          cell[r * (SW + 1)] = ds[r];                                    // W = skew(dS) replaces U in place
        }
        const S3 pps = split2h1(pp[0], pp[1], pp[2], pp[3], one), dss = splitn<true>(ds);
#endif
        // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]: the accumulator registers ARE the B operands
        prodx2<true>(docol, pps, dv[s], qcol, dss, dk[s]);
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: both operands through hardware-transposed reads (product: next step)
        st_planes<true>(Dst, 512, dss);
        dst_p = tr_planes<true>(Dtr, 512);
#pragma unroll
        for (int r = 0; r < 4; ++r) cu[r] = cn_[r];
      }
      dq = prodx<true>(kcol_p, dst_p, dq);
    } else {
      // generic form: a wave-uniform branch per step
#pragma unroll
      for (int s = 0; s < CN; ++s) {
        if (s < cnt) {                                                   // wave-uniform
          const int kt = kt0 + s;
          const S3 krow = ld_planes<true>(Krow0 + s * 1024, 512);
          f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          prodx2<true>(qrow, krow, s4, dorow, vfr[s], dp);
          f32x4 pp, ds;
          const bool tail = kt == nkt - 1;
          const bool kv = kt * 16 + c < n;
          float* cell = cellP + 16 * (KPW - 1 - s);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(s4[r] + cell[r * (SW + 1)], sc2, nlse2[r]));
            if (tail) p = kv ? p : 0.f;
            pp[r] = p;
            ds[r] = p * fmaf(dp[r], kD, -dl4s[r]);
            cell[r * (SW + 1)] = ds[r];
          }
          const S3 pps = split2h1(pp[0], pp[1], pp[2], pp[3], one), dss = splitn<true>(ds);
          prodx2<true>(docol, pps, dv[s], qcol, dss, dk[s]);
          st_planes<true>(Dst, 512, dss);
          const S3 kcol = tr_planes<true>(Kcol0 + s * 1024, 512);
          const S3 dst = tr_planes<true>(Dtr, 512);
          dq = prodx<true>(kcol, dst, dq);
        }
      }
    }
    stamp(qt, 1);
    if (loader) stage_store();              // (every wave took its fragments of THIS tile before the previous barrier)
    stamp(qt, 2);
    __syncthreads();                                                     // Ba: every cell of sp holds dS; the stage holds tile qt + 1

    // ================= P2 =================
    stamp(qt, 3);
    // (the E row fragments of the next tile's positions are requested here and land while the strip is consumed)
    S3 es[2 * NCA];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
      for (int t = 0; t < 2; ++t) es[2 * ci + t] = e_rows(more ? qt + 1 : qt, jn[ci] + M * t, own(ci, jn[ci] + M * t));
    }
    // ---- consume the strip: dQ^T += E^T W^T, dE^T += Q^T W for the tiles this wave owns (all W reads first: they may alias
    // the transposition image as far as the compiler knows, and would otherwise wait behind each tile's write) ----
    if (!(a.dbg & 4)) {
      float4 w4s[2 * NCA];
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = j0[ci] + M * t;
          float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (own(ci, j)) {                                              // wave-uniform (XT: no test)
            w4 = *reinterpret_cast<const float4*>(sp + c * SW + 16 * (NKTM - j) + 4 * g);         // W[a = c][dl = 4g + i]
            // cell (a, dl) of position j belongs to key 16 j + a - dl: the cells of keys outside the sequence still hold U
            const bool e0 = t == 0 && j == 0, en = t == 1 && j == nkt;   // (M <= nkt: position nkt is always an older tile)
            w4.x = ((e0 && z0[0]) || (en && zn[0])) ? 0.f : w4.x;
            w4.y = ((e0 && z0[1]) || (en && zn[1])) ? 0.f : w4.y;
            w4.z = ((e0 && z0[2]) || (en && zn[2])) ? 0.f : w4.z;
            w4.w = ((e0 && z0[3]) || (en && zn[3])) ? 0.f : w4.w;
          }
          w4s[2 * ci + t] = w4;
        }
      }
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = j0[ci] + M * t;
          if (own(ci, j)) {
            const float4 w4 = w4s[2 * ci + t];
#if SE_ATTN_TWIN == 1
            const S3 ws = twin_raw_(w4.x, w4.y, w4.z, w4.w);
#else
            const S3 ws = split2h1(w4.x, w4.y, w4.z, w4.w, one);          // (W = skew(dS): already at dS's scale)
#endif
            st_planes<true>(Dst, 512, ws);                               // image [a][dl]
            const S3 wt = tr_planes<true>(Dtr, 512);                     // W[a = 4g + i][dl = c]
            // dQ^T[d][q] += E^T[d][dl] W^T[dl][q];  dE^T[d][dl] += Q^T[d][q] W[q][dl]
            if (t == 0) prodx2<true>(ecs[2 * ci + t], ws, dq2, qcol, wt, de0[ci]);
            else prodx2<true>(ecs[2 * ci + t], ws, dq3, qcol, wt, de1[ci]);
          }
        }
      }
    }
    stamp(qt, 4);
    // ---- U of the next query tile into the other strip ----
    S3 qrow_n = qrow;
    if (more) qrow_n = ld_planes<true>(Qimg + rfo, 512);
    if (more && !(a.dbg & 1)) {
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = jn[ci] + M * t;
          if (own(ci, j)) u_tile(spn, qrow_n, es[2 * ci + t], j);        // wave-uniform
        }
      }
    }
    // ---- this wave's part of dQ; the fragments of the next tile ----
#pragma unroll
    for (int r = 0; r < 4; ++r) dq[r] = fmaf(dq[r], cq1, (dq2[r] + dq3[r]) * cq2);
    *reinterpret_cast<float4*>(dqs + wave * 256 + c * 16 + 4 * g) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    if (more) { qrow = qrow_n; read_frags(qt + 1, false); }
    stamp(qt, 5);
    __syncthreads();                                                     // Bb: sp consumed, spn written, partials written
    stamp(qt, 6);
    if (wave == qt % NW && q0 + c < n) {
      float4 t0 = *reinterpret_cast<const float4*>(dqs + c * 16 + 4 * g);
#pragma unroll
      for (int w = 1; w < NW; ++w) {                                     // fixed order: deterministic
        const float4 t1 = *reinterpret_cast<const float4*>(dqs + w * 256 + c * 16 + 4 * g);
        t0.x += t1.x; t0.y += t1.y; t0.z += t1.z; t0.w += t1.w;
      }
      *reinterpret_cast<float4*>(dqb + (unsigned)((q0 + c) * ps * 192 + 4 * g)) = t0;
      omax = fmaxf(fmaxf(omax, fmaxf(fabsf(t0.x), fabsf(t0.y))), fmaxf(fabsf(t0.z), fabsf(t0.w)));
    }
    // ---- the windows of the classes advance: the tile at position nkt has had its last contribution ----
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
      if (XT || ci < ncls) {
        if (j0[ci] + M == nkt) flush(de1[ci], qt - nkt);                // (M <= nkt: position nkt is always a class's older tile)
        if (jn[ci] == 0) {
          de1[ci] = de0[ci];
          de0[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      j0[ci] = jn[ci];
    }
    stamp(qt, 7);
  }
  // the tiles still in the windows (as if at query tile nqt): positions 1 .. nkt hold the offset tiles nqt - j
#pragma unroll
  for (int ci = 0; ci < NC; ++ci) {
    if (XT || ci < ncls) {
      if (j0[ci] >= 1) flush(de0[ci], nqt - j0[ci]);
      if (j0[ci] + M <= nkt) flush(de1[ci], nqt - j0[ci] - M);
    }
  }
  // ---- dK, dV of the wave's keys: C layout [d = 4g + r][key = c] ----
#pragma unroll
  for (int s = 0; s < CN; ++s) {
    const int key = (kt0 + s) * 16 + c;
    if (s < cnt && key < n) {
      float* p = dqb + (unsigned)(key * ps * 192 + 4 * g);
      *reinterpret_cast<float4*>(p + 64) = make_float4(dk[s][0] * cq1, dk[s][1] * cq1, dk[s][2] * cq1, dk[s][3] * cq1);
      *reinterpret_cast<float4*>(p + 128) = make_float4(dv[s][0] * cdv, dv[s][1] * cdv, dv[s][2] * cdv, dv[s][3] * cdv);
#pragma unroll
      for (int r = 0; r < 4; ++r) omax = fmaxf(omax, fmaxf(fabsf(dk[s][r] * cq1), fabsf(dv[s][r] * cdv)));
    }
  }
  if (a.dqkv_amax) {
    omax = wave_max(omax);
    if (lane == 0) amax_raise_(a.dqkv_amax, omax);
  }
}

// MINW: waves per SIMD the register allocation is bounded for.  Exact bodies are named by the instantiation: (CA key tiles, NA
// classes) for the waves before the last one, (CB, NB) with the key mask of the tail tile (TL) for the last wave, optionally a
// second pair (CC, NC_) for earlier waves (CC < 0: none); every other split takes the tested body.  n = 321: four waves, key tiles
// 5 + 5 + 5 + 6, classes 3 + 3 + 3 + 2 -> A = (5, 3), B = (6, 2).  n = 101: two waves, 3 + 4 key tiles, 2 + 2 classes -> A = (3, 2),
// B = (4, 2) (four waves with 1 + 2 + 2 + 2 key tiles: 0.79 vs 0.77 ms).
// BARRIER CONTRACT: the waves of one workgroup return into DIFFERENT attn_bwd4_body instantiations (XT / TL variants), each with its
// own __syncthreads() sites; s_barrier counts arrivals regardless of the program counter, so this is correct exactly as long as
// every variant executes the same number of barriers: 2 before the query-tile loop + 2 per query tile, none behind a dbg switch.
// tests/test_attn_gpu.py::test_bwd4_generic_body_beside_exact_bodies runs the generic body (SE_ATTN_DBG=64) against the exact
// bodies on the same shapes: a barrier mismatch shows up there as a hang (the test runs under a timeout), not in training.
// GEN = false: the exact bodies ONLY -- the host launches this form exactly when every wave of the plan matches one (attn_bwd4_exact:
// the same predicate).  GEN = true: the generic body for every wave (every other split; SE_ATTN_DBG=64 forces it).  Two kernels
// instead of one with both: the generic body's register demand (KPW key tiles AND NCW classes in one wave) gave the merged kernel
// 76 bytes of scratch per lane although the default path never ran the spilling code (round-4 review) -- a kernel with scratch
// pays for the allocation at every wave launch.
template <int NW, int CA, int NA, int CB, int NB, int CC, int NC_>
static inline bool attn_bwd4_exact(const AttnBwd4Plan& pl, int nkt, int dbg) {
  if (2 * pl.M != nkt + 1 || (dbg & 64)) return false;
  for (int w = 0; w < NW; ++w) {
    const int cnt = pl.cnt[w], ncls = pl.ncls[w];
    const bool ok = w == NW - 1 ? (cnt == CB && ncls == NB) : ((cnt == CA && ncls == NA) || (CC >= 0 && cnt == CC && ncls == NC_));
    if (!ok) return false;
  }
  return true;
}
template <int NW, int KPW, int NCW, int NKTM, int MINW, int CA, int NA, int CB, int NB, int CC = -1, int NC_ = 0, bool GEN = false>
// (GEN: one wave per SIMD bounds the allocation at 512 registers -- the generic body holds KPW key tiles AND NCW classes in one wave
// and spilled 56 bytes per lane under the two-wave bound; a fallback form may be slower, not carry scratch)
__global__ __launch_bounds__(NW * 64, GEN ? 1 : MINW) void attn_bwd4_kernel(AttnBwd4Args a, AttnBwd4Plan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (GEN) {
    attn_bwd4_body<NW, KPW, NCW, NKTM, KPW, NCW, false, true>(a, pl, smem4, wave, lane);
  } else {
    const int cnt = pl.cnt[wave], ncls = pl.ncls[wave];
    const bool last = wave == NW - 1;
    if (last) return attn_bwd4_body<NW, KPW, NCW, NKTM, CB, NB, true, true>(a, pl, smem4, wave, lane);
    if constexpr (CC >= 0) {
      if (cnt == CC && ncls == NC_) return attn_bwd4_body<NW, KPW, NCW, NKTM, (CC >= 0 ? CC : 1), NC_, true, false>(a, pl, smem4, wave, lane);
    }
    (void)cnt; (void)ncls;
    attn_bwd4_body<NW, KPW, NCW, NKTM, CA, NA, true, false>(a, pl, smem4, wave, lane);      // (host-checked: attn_bwd4_exact)
  }
}
