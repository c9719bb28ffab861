// bf16 MFMA GEMM, second kernel: 128x256x64 tile, THREE LDS stages, register-direct
// epilogue (gfx950).   C[M,N] = epilogue(A[M,K] @ W[N,K]^T), nn.Linear layout.
// Reference call sites: models/dit.py:277-287 (out-projection / MLP-down + gate *
// x + residual), 225-228, 256, 306/320.
//
// Why a second kernel.  In gemm_bf16.hip a CU alternates an MFMA phase (main loop)
// and an HBM phase (epilogue): one 256x256 tile's 128 accumulators per lane leave
// no registers for anything to run beside them.  Here a wave owns 64x64 of a
// 128x256 tile = 64 accumulators, so TWO sets fit: while set B accumulates tile
// i+1, set A's finished tile leaves from registers in small units placed between
// the MFMA groups of the K loop -- no LDS transposition (a lane holds 4 consecutive
// fp32 columns = one 16-byte store; for bf16 outputs the W rows of the LDS image
// are permuted on the DMA source side so that a lane holds 8 consecutive columns),
// no barrier coupling the waves at a tile end.  The K slabs of all tiles of a
// workgroup form one stream through a 3-stage LDS ring filled by LDS-DMA two slabs
// ahead (counted vmcnt, raw s_barrier).
//
// 8 waves = 2 (M) x 4 (N); MFMA 16x16x32 bf16 issued as W_frag x A_frag (C^T tiles:
// a lane holds 4 consecutive output columns of one row), same LDS image and swizzle
// as gemm_bf16.hip, same K order per output row => bit-identical results.
#include "gemm_common.h"

namespace mdlm {
namespace pp {

constexpr int TBM = 128, TBN = 256, NW = 8;
constexpr int A_BYTES = TBM * BK * 2;       // 16 KiB
constexpr int B_BYTES = TBN * BK * 2;       // 32 KiB
constexpr int STAGE = A_BYTES + B_BYTES;    // 48 KiB
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE + 16384;   // 144 KiB ring + column constants (gate, bias: N <= 2048)
constexpr int A_CH = (TBM / 8) / NW;        // 2 DMA instructions (1 KiB each) per wave and slab
constexpr int B_CH = (TBN / 8) / NW;        // 4
constexpr int GROUP_M = 1024 / TBM;         // M tiles per A-row group of the tile walk

#define PP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// One slab step = stage slab + 2, 32 MFMAs on the slab the ring holds, wait for slab + 1.
// A UNIT slab step also lets 16 rows x 64 columns (unit U) of the PREVIOUS tile -- the
// other accumulator set -- leave: its 4 residual loads are issued in FRONT of the slab's
// DMA, so the ordinary end-of-slab wait (all but the 6 youngest operations) covers them;
// then the values are formed and stored.  Vector-memory operations complete in issue order
// (loads, stores and LDS-DMA on one counter), so the slab behind a unit slab waits with the
// 4 stores counted (10).  Nothing is held in registers across a slab boundary.
enum { PLAIN = 0, EPI_SLAB = 1 };

template <int EPI, typename TC>
__global__ __launch_bounds__(512, 2)
void gemm_pp_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int fq = lane >> 4, fr = lane & 15;

  // ---- XCD-aware persistent tile walk (as gemm_bf16.hip, 128-row tiles) ----------
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int slots = gridDim.x >> 3;
  const int per_group = GROUP_M * g.tiles_n;
  const int64_t m_lim = g.m_rows_dev ? (int64_t)*g.m_rows_dev : g.M;
  const int64_t vm_live = (m_lim + TBM - 1) / TBM;
  const int vm = vm_live < g.tiles_m ? (int)vm_live : g.tiles_m;
  const int total = vm * g.tiles_n;
  const int per_xcd = (total + 7) >> 3;
  const int s_begin = xcd * per_xcd;
  const int s_end = total < s_begin + per_xcd ? total : s_begin + per_xcd;
  const int cnt = s_end > s_begin ? s_end - s_begin : 0;
  auto tile_at = [&](int j, int& tm, int& tn) -> bool {
    if (j >= cnt) return false;
    const int s = s_begin + j;
    const int gl = s / per_group;
    const int rem = s - gl * per_group;
    const int left = vm - gl * GROUP_M;
    const int gm = left < GROUP_M ? left : GROUP_M;
    tn = rem / gm;
    tm = gl * GROUP_M + (rem - tn * gm);
    return true;
  };
  int j = bid >> 3, tm, tn;
  if (!tile_at(j, tm, tn)) return;
  const int KT = g.K / BK;

  // ---- column constants (gate, bias) of ALL N columns -> LDS behind the ring -------
  // record c/4 = {gate[c..c+3], bias[c..c+3]} (32 bytes)
  float* const lds_cc = reinterpret_cast<float*>(smem + NSTAGE * STAGE);
  if constexpr (EPI == MDLM_EPI_GATE_RESIDUAL) {
    for (int i = tid; i < g.N; i += 512) {
      lds_cc[(i >> 2) * 8 + (i & 3)] = g.gate[i];
      lds_cc[(i >> 2) * 8 + 4 + (i & 3)] = g.bias ? g.bias[i] : 0.f;
    }
    __syncthreads();
  } else {   // bf16 outputs: bias[N] as it stands
    for (int i = tid; i < g.N; i += 512) lds_cc[i] = g.bias ? g.bias[i] : 0.f;
    __syncthreads();
  }

  // ---- loader: the slab stream, two slabs ahead of the MFMAs -----------------------
  // DMA source = wave-uniform base (tile, chunk, K step: scalar registers) + a 32-bit
  // lane offset (row inside the 8-row chunk group, swizzled 16-byte chunk): M % 128 == 0
  // and N % 256 == 0, so no row is ever clamped.
  const int srow = lane >> 3;
  const int skc = ((lane & 7) ^ srow) * 8;
  const int a_voff = (int)(((int64_t)(wid * 8 + srow) * g.lda + skc) * 2);
  // bf16 outputs: LDS row (nu, i) of a wave's 64 W rows (nu = 16-row MFMA block, i = row in
  // it) holds W row 32 (nu / 2) + 8 (i / 4) + 4 (nu % 2) + i % 4, so that after the MFMAs a
  // lane's values of blocks 2p and 2p + 1 are 8 CONSECUTIVE output columns: one 16-byte store
  constexpr bool BF16_OUT = sizeof(TC) == 2;
  const int w_lds = wid * 8 + srow;   // LDS row inside the 64-row group this wave stages
  const int w_row = BF16_OUT ? 32 * (w_lds >> 5) + 8 * ((w_lds & 15) >> 2) + 4 * ((w_lds >> 4) & 1) + (w_lds & 3)
                             : w_lds;
  const int b_voff = (int)(((int64_t)w_row * g.ldw + skc) * 2);
  const char* a_tile = nullptr;   // first row of the loader's tile in A / W (uniform)
  const char* b_tile = nullptr;
  auto set_src = [&](int tm_, int tn_) {
    a_tile = reinterpret_cast<const char*>(g.A + (int64_t)tm_ * TBM * g.lda);
    b_tile = reinterpret_cast<const char*>(g.W + (int64_t)tn_ * TBN * g.ldw);
  };
  int jL = j, ktL = 0;       // next slab to stage: tile jL, K step ktL
  int ls = 0;                // LDS offset of the stage it goes to
  set_src(tm, tn);
  // Stage the next slab of the stream.  Past the end of the stream the last slab is
  // staged again (into a stage nobody reads any more), so that every slab step
  // issues the same number of vector-memory operations and the counted waits hold.
  auto stage_next = [&]() {
    char* const sb = smem + ls;
    const int64_t k0 = (int64_t)ktL * BK * 2;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      GLDS16(a_tile + (k0 + (int64_t)i * 64 * g.lda * 2) + a_voff, sb + (i * NW + wid) * 1024);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      GLDS16(b_tile + (k0 + (int64_t)i * 64 * g.ldw * 2) + b_voff, sb + A_BYTES + (i * NW + wid) * 1024);
    ls = ls == (NSTAGE - 1) * STAGE ? 0 : ls + STAGE;
    if (jL >= 0 && ++ktL == KT) {
      int tmL, tnL;
      if (tile_at(jL + slots, tmL, tnL)) { jL += slots; ktL = 0; set_src(tmL, tnL); }
      else { jL = -1; ktL = KT - 1; }
    }
  };

  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    foff[ks] = fr * 128 + (((ks * 4 + fq) ^ (lane & 7)) * 16);
  const int a_base = wr * 64 * 128;
  const int b_base = A_BYTES + wc * 64 * 128;

  f32x4 acc[2][4][4];
  f32x4 rs[4] = {};                    // residual rows of the unit that leaves next
  int64_t d_row0 = 0; int d_col0 = 0;  // wave tile origin of the tile that is draining
  float* const Cp = reinterpret_cast<float*>(g.C);

  // Residual of unit u of the wave tile at (row0, col0): a lane's 4 x 16 bytes.  Inline
  // asm on purpose: beside LDS-DMA in flight hipcc waits vmcnt(0) for any ordinary load
  // (it cannot tell the DMA's LDS writes from the load), which would drain the slab ring at
  // every unit; the counted wait is ours (wait_unit).
  auto load_unit = [&](int64_t row0, int col0, int u) {
    const float* rp = g.residual + (row0 + u * 16 + fr) * g.ldc + col0 + fq * 4;
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:64\n\t"
                 "global_load_dwordx4 %2, %4, off offset:128\n\t"
                 "global_load_dwordx4 %3, %4, off offset:192"
                 : "=&v"(rs[0]), "=&v"(rs[1]), "=&v"(rs[2]), "=&v"(rs[3]) : "v"(rp) : "memory");
  };
  // residual + gate * (acc + bias) of unit U of set P (dit.py:28-34, eval), in place in `rs`.
  // The column constants come from LDS by inline asm for the same reason (an ordinary
  // ds_read beside LDS-DMA in flight gets a vmcnt(0) in front of it).
  auto compute_unit = [&](auto p_c, auto u_c) {
    constexpr int P = decltype(p_c)::value, U = decltype(u_c)::value;
    const unsigned cc = (unsigned)(NSTAGE * STAGE) + (unsigned)(d_col0 + fq * 4) * 8u;
    f32x4 gt[4], b4[4];
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %8 offset:16\n\t"
                 "ds_read_b128 %2, %8 offset:128\n\t"
                 "ds_read_b128 %3, %8 offset:144\n\t"
                 "ds_read_b128 %4, %8 offset:256\n\t"
                 "ds_read_b128 %5, %8 offset:272\n\t"
                 "ds_read_b128 %6, %8 offset:384\n\t"
                 "ds_read_b128 %7, %8 offset:400\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(gt[0]), "=&v"(b4[0]), "=&v"(gt[1]), "=&v"(b4[1]),
                   "=&v"(gt[2]), "=&v"(b4[2]), "=&v"(gt[3]), "=&v"(b4[3])
                 : "v"(cc) : "memory");
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4 a = acc[P][U][n];
#pragma unroll
      for (int k = 0; k < 4; ++k) rs[n][k] = rs[n][k] + gt[n][k] * (a[k] + b4[n][k]);
    }
  };
  auto store_rs = [&](int u) {
    float* cp = Cp + (d_row0 + u * 16 + fr) * g.ldc + d_col0 + fq * 4;
    const int64_t s1 = 16, s2 = 32;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      store16<false>(cp + (n & 1) * s1 + (n >> 1) * s2, make_uint4(__float_as_uint(rs[n][0]), __float_as_uint(rs[n][1]), __float_as_uint(rs[n][2]), __float_as_uint(rs[n][3])), GEMM_DBG(g));
  };

  // bf16 outputs: unit U of set P = 16 rows x 64 columns; a lane's two 16-byte stores
  // (8 consecutive columns each, see w_row): gelu_tanh(acc + bias) (dit.py:225-228)
  auto bf16_unit = [&](auto p_c, auto u_c) {
    constexpr int P = decltype(p_c)::value, U = decltype(u_c)::value;
    const unsigned cc = (unsigned)(NSTAGE * STAGE) + (unsigned)(d_col0 + fq * 8) * 4u;
    f32x4 b[4];
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:128\n\t"
                 "ds_read_b128 %3, %4 offset:144\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(cc) : "memory");
    bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (d_row0 + U * 16 + fr) * g.ldc + d_col0 + fq * 8;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      unsigned o[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 a = acc[P][U][2 * pr + h];
        const f32x4 bb = b[2 * pr + h];
        f32x2 lo = {a[0] + bb[0], a[1] + bb[1]};
        f32x2 hi = {a[2] + bb[2], a[3] + bb[3]};
        if constexpr (EPI == MDLM_EPI_BIAS_GELU) { lo = gelu_tanh2(lo); hi = gelu_tanh2(hi); }
        o[2 * h] = pack_bf16x2(lo.x, lo.y);
        o[2 * h + 1] = pack_bf16x2(hi.x, hi.y);
      }
      store16<true>(cp + pr * 32, make_uint4(o[0], o[1], o[2], o[3]), GEMM_DBG(g));
    }
  };

  int cs = 0;                          // LDS offset of the stage the MFMAs read
  int64_t grow0; int gcol0;            // wave tile origin of the tile in progress

  bool after_unit = false;             // the previous slab step of this wave issued 4 stores behind its DMA
  auto slab = [&](auto s_c, auto kind_c, auto u_c) {
    constexpr int S = decltype(s_c)::value, KIND = decltype(kind_c)::value;
    constexpr int U = decltype(u_c)::value;
    if constexpr (KIND == EPI_SLAB && !BF16_OUT) load_unit(d_row0, d_col0, U);
    stage_next();                                        // slab + 2

    const char* At = smem + cs + a_base;
    const char* Bt = smem + cs + b_base;
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        bfr[ks][n] = *reinterpret_cast<const bf16x8*>(Bt + n * 2048 + foff[ks]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        af[ks][m] = *reinterpret_cast<const bf16x8*>(At + m * 2048 + foff[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[S][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][n], af[ks][m], acc[S][m][n], 0, 0, 0);
    // 8 reads up front, then one behind every pair of MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    // slab + 1 landed; slab + 2 (6) and, behind a unit slab, its 4 stores stay in flight
    if constexpr (BF16_OUT) { if (after_unit) PP_WAIT_VM(8); else PP_WAIT_VM(6); }
    else { if (after_unit) PP_WAIT_VM(10); else PP_WAIT_VM(6); }
    after_unit = KIND == EPI_SLAB;
    if constexpr (KIND == EPI_SLAB) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BF16_OUT) {
        bf16_unit(std::integral_constant<int, 1 - S>{}, u_c);
      } else {
        compute_unit(std::integral_constant<int, 1 - S>{}, u_c);
        store_rs(U);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cs = cs == (NSTAGE - 1) * STAGE ? 0 : cs + STAGE;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using KPlain = std::integral_constant<int, PLAIN>;
  using KEpi = std::integral_constant<int, EPI_SLAB>;

  // One tile into set S.  HAS_PREV: the previous tile (other set) drains in 4 units, at
  // slab steps ph, ph + gap, ph + 2 gap, ph + 3 gap of the K loop (gap = KT / 4 >= 2, so a
  // unit slab is never directly behind another one).  The phase ph differs from wave to
  // wave and from workgroup to workgroup, so that the units' HBM traffic is spread evenly
  // over time instead of arriving as one burst of all waves of all CUs.
  const int gap = KT >> 2;
  const int ph = (wid + (bid >> 3)) % gap;
  auto tile = [&](auto s_c, auto has_prev_c) {
    constexpr int S = decltype(s_c)::value;
    constexpr bool HAS_PREV = decltype(has_prev_c)::value;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[S][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (HAS_PREV) {
      // straight-line unit slabs between loops of plain ones (a switch on the unit index
      // inside one K loop makes hipcc spill the accumulator sets)
      for (int i = 0; i < ph; ++i) slab(s_c, KPlain{}, I0{});
      slab(s_c, KEpi{}, I0{});
      for (int i = 1; i < gap; ++i) slab(s_c, KPlain{}, I0{});
      slab(s_c, KEpi{}, I1{});
      for (int i = 1; i < gap; ++i) slab(s_c, KPlain{}, I0{});
      slab(s_c, KEpi{}, I2{});
      for (int i = 1; i < gap; ++i) slab(s_c, KPlain{}, I0{});
      slab(s_c, KEpi{}, I3{});
      for (int i = ph + 3 * gap + 1; i < KT; ++i) slab(s_c, KPlain{}, I0{});
    } else {
      for (int kt = 0; kt < KT; ++kt) slab(s_c, KPlain{}, I0{});
    }
  };

  stage_next();
  stage_next();
  PP_WAIT_VM(6);                       // slab 0 landed (this wave's part), slab 1 in flight
  __builtin_amdgcn_s_barrier();

  auto next_tile = [&]() -> bool {
    d_row0 = grow0; d_col0 = gcol0;    // the tile just finished drains next
    j += slots;
    if (!tile_at(j, tm, tn)) return false;
    grow0 = (int64_t)tm * TBM + wr * 64; gcol0 = tn * TBN + wc * 64;
    return true;
  };
  grow0 = (int64_t)tm * TBM + wr * 64; gcol0 = tn * TBN + wc * 64;
  tile(I0{}, std::false_type{});
  int last = 0;                        // set holding the tile that has not left yet
  while (next_tile()) {
    tile(I1{}, std::true_type{});
    last = 1;
    if (!next_tile()) break;
    tile(I0{}, std::true_type{});
    last = 0;
  }
  // ---- the last tile leaves (unit 0's residual is in flight already) ----------------
  if (!(GEMM_DBG(g) & 2)) {
    auto drain = [&](auto p_c) {
      if constexpr (BF16_OUT) {
        bf16_unit(p_c, I0{}); bf16_unit(p_c, I1{}); bf16_unit(p_c, I2{}); bf16_unit(p_c, I3{});
        return;
      }
      load_unit(d_row0, d_col0, 0); PP_WAIT_VM(0); compute_unit(p_c, I0{}); store_rs(0);
      load_unit(d_row0, d_col0, 1); PP_WAIT_VM(0); compute_unit(p_c, I1{}); store_rs(1);
      load_unit(d_row0, d_col0, 2); PP_WAIT_VM(0); compute_unit(p_c, I2{}); store_rs(2);
      load_unit(d_row0, d_col0, 3); PP_WAIT_VM(0); compute_unit(p_c, I3{}); store_rs(3);
    };
    if (last == 0) drain(I0{}); else drain(I1{});
  }
  PP_WAIT_VM(0);   // the two slabs staged past the end of the stream
}

template <int EPI, typename TC>
static int launch_pp(GemmArgs g, hipStream_t s) {
  static bool attr_done[MAX_DEVICES] = {};
  int dev = 0;
  if (int rc = ensure_dynamic_lds(&gemm_pp_kernel<EPI, TC>, LDS_BYTES, attr_done, "gemm_pp", &dev)) return rc;
  g.tiles_m = (int)(g.M / TBM);
  g.tiles_n = (g.N + TBN - 1) / TBN;
  int64_t grid = (((int64_t)g.tiles_m * g.tiles_n + 7) / 8) * 8;
  if (grid > device_cus(dev)) grid = device_cus(dev);
  gemm_pp_kernel<EPI, TC><<<(unsigned)grid, 512, LDS_BYTES, s>>>(g);
  MDLM_LAUNCH_CHECK();
  return 0;
}

}  // namespace pp

// Entry used by mdlm_gemm_bf16 (gemm_bf16.hip) when the shape / epilogue is one this
// kernel covers: returns -2 when it does not apply.
int gemm_pp_dispatch(const GemmArgs& g, int epilogue, int dtype_c, hipStream_t s) {
  if (g.M % pp::TBM || g.K % BK || g.N % 64 || !g.vec_ok) return -2;
  // one gate row for every output row (sigma zeroed: MDLM sampling / evaluation), all N
  // columns' constants in LDS, at least three slab steps per quarter of the K loop (KT >= 12)
  if (epilogue == MDLM_EPI_GATE_RESIDUAL && dtype_c == MDLM_F32 && g.N % pp::TBN == 0 &&
      g.N <= 2048 && g.n_mod == 1 && g.K >= 8 * BK)
    return pp::launch_pp<MDLM_EPI_GATE_RESIDUAL, float>(g, s);
  if (epilogue == MDLM_EPI_BIAS_GELU && dtype_c == MDLM_BF16 && g.N % pp::TBN == 0 && g.N <= 4096 &&
      g.K >= 8 * BK)
    return pp::launch_pp<MDLM_EPI_BIAS_GELU, bf16_t>(g, s);
  return -2;
}

}  // namespace mdlm
