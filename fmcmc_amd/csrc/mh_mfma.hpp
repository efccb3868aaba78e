// mh_mfma.hpp -- mh_sweep_mfma<KIND, NG, NS, DBG>: the headline kernel, bit-exact fp64 MFMA evaluation.
#pragma once

namespace {

// ==============================================================================================
// MFMA evaluation kernel: Gaussian linear regression whose data fit the operand registers (p <= 3, n <= 10240 with one
// group of K = 4; p <= 7, n <= 5120 with two chained groups).  Described below for p = 3.
//
// v_mfma_f64_4x4x4_4b_f64 on gfx950 (measured, tools/mfma64_exact.hip): k = lane / 16; inside a 16-lane group the
// A operand sits at 4*blk + i, B at 4*blk + j, the result D at lane 16*i + 4*blk + j; and it is BITWISE the chain
//     D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))).
// With A = [x1, x2, x3, y], B = [b1, b2, b3, -1] and C = b0 this is exactly the canonical
//     m = fma(x3, b3, fma(x2, b2, fma(x1, b1, b0)));   -r = fma(y, -1, m)
// for 16 observations x 4 chains per instruction, and fma(D, D, acc) == fma(r, r, acc) bit for bit.
// One MFMA replaces 16 x 4 x 4 = 256 VALU lane-FMAs + 64 subtractions with ONE issue slot, which lifts the
// fp64 pipe out of the VALU issue limit (~6.2 ticks per instruction at 2 waves per SIMD).
//
// Mapping that keeps the canonical reduction: wave w owns canonical lanes 64w..64w+63; MFMA t = 4*s + g covers
// slot s of the canonical lanes 64w + 16*i_row + 4*blk + g (operand position o = 4*blk + i_row); result lane L holds
// chain j = L % 4 of canonical lane 64w + 16*(L/16) + 4*((L/4)%4) + g, accumulated in acc[g] in slot order.  The bits of
// a canonical lane index are laid out so that the xor-butterfly tree costs next to nothing: levels 1, 2 pair the four
// ACCUMULATORS of a lane (three adds, no cross-lane traffic), levels 4, 8 the four blocks of a 16-lane row (two DPP row
// shifts), levels 16, 32 the four rows (permlane16 / permlane32 swaps): 15 instructions per wave BEFORE the barrier leave
// one double per chain and wave, and the owners finish levels 64..256 over 8 values behind it.  (Until round 2 the lanes
// wrote all 4 x 512 partials to a transposed LDS tile and the owners folded them behind the barrier: 810 of the 1600 ticks
// of the exposed owner phase.  Same tree, same bits.)
// All four chains of the workgroup are evaluated together, so this kernel is not chain-pipelined: a step is
// evaluation + in-wave fold | barrier | 4 owner phases in parallel (waves 0..3, priority raised) | barrier.
// ==============================================================================================
constexpr int MF_NMF = 80;   // MFMAs per wave per step: 20 slots x 4 lane groups
// chain stride of the partial tiles: here one ds_write_b64 carries 4 chains x 16 canonical lanes; with the row stride
// 66 the 16 lanes of a write group land on double-banks {0,8,1,9} + 2*(l&7), so a chain stride == 2 (mod 16) spreads the
// four chains over all 16 double-banks (8*66 = 528 == 0 mod 16 made every write a 4-way conflict)
constexpr int MF_TCS = 8 * PIPE_TRS + 2;   // (the replicated kernel's tile, mh_mfma_rep.hpp)

// DBG is a TEMPLATE parameter on purpose: the MFMA loop is sensitive to every live register (fewer free VGPRs = fewer
// MFMA results in flight before their dependent fma); stamp code that is merely disabled at run time cost 13 %.
// Shapes: NG groups of K = 4 carry up to 4*NG - 1 covariates plus y (unused k-slots are zero: fma(0, 0, acc) == acc
// exactly), the 80 operand registers of a lane hold 20 / NG observation slots, i.e. NG = 1: p <= 3, n <= 10240;
// NG = 2: p <= 7, n <= 5120.  The number of observation slots NS = ceil(n / 512) is a TEMPLATE parameter: with a run-time
// count every batch of MFMAs becomes a basic block, the scheduler can no longer overlap the FMAs of one batch with the
// MFMAs of the next, and the step is 22 % slower (measured).  Only the last slot holds padding and pays for masks.
// BIG: offsets of rows and variates are 32 bits from the CHAIN's own blocks (64-bit wave-uniform bases computed here) instead
// of 32 bits from the buffer bases (kernel arguments): the form for calls with more than 4 GiB of samples or of fed
// variates.  A template parameter because the extra scalar registers cost the headline shape 1.7 % (the kernel sits at
// the SGPR limit); the dispatcher takes BIG only where the sizes need it.
// EXT (round 4): data sets beyond the operand registers (n > 10240 at p <= 3, n > 5120 at p <= 7).  NS observation slots
// stay resident as before; the remaining A.mf_next slots are STREAMED every step from A.mf_stream, a copy of the data in
// operand order (mfma_build_stream: [wave][slot][group][lane][4 lane-group values], 32 contiguous bytes per lane, so a slot is
// two global_load_dwordx4 per lane and group, fully coalesced, L2-resident and shared by all workgroups), RD slots ahead in
// a register ring.  Same MFMA chain, same accumulators in slot order: the canonical bits.  The streamed part runs at what a
// CU gets out of its L2 (~16 KB per slot and workgroup), so the rate per flop falls off gently with n instead of dropping to
// the general kernel's at observation 10,241 (profiles/r04_shape_map.md).
template <int KIND, int NG, int NS, bool DBG, bool BIG = false, bool EXT = false>
__global__ __launch_bounds__(NT) void mh_sweep_mfma(const SweepArgs A) {
  constexpr int CW = 4;
#ifndef MFO_MB
#define MFO_MB 12
#endif
  // (slot, lane group) pairs per batch: 8 until round 3; 12 is 1.9 % faster per step (tools/probe_mfma4_pattern.hip: a pair costs
  //  21.5 / 21.0 cycles at 8 / 16 per batch), 16 another 0.5 % but spills six registers
  constexpr int MB = MFO_MB;
  constexpr int TN = NS * 4;        // pairs held per lane and group (NG * TN <= MF_NMF registers)
  static_assert(NG * TN <= MF_NMF, "operand registers");
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX]
  double* s_par = s_th1 + CW * PIPE_KMAX;          // [4][PIPE_KMAX]
  double* s_fold = s_par + 4 * PIPE_KMAX;          // [CW][NW] per-wave sums of every chain (canonical levels 1..32 done)
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;

  // ---- A operand: feature kf_ = lane / 16 of the observation of canonical lane 64 w + 16 i_row + 4 blk + g in slot s, for
  //      t = 4 s + g and operand position lane % 16 = 4 blk + i_row
  const int feat = lane >> 4, o16 = lane & 15;
  const int cl_a = 16 * (o16 & 3) + 4 * (o16 >> 2);
  const int P = A.p;
  double areg[NG][TN];
#pragma unroll
  for (int q = 0; q < NG; q++) {
    const int f = 4 * q + feat;   // column of [x_1 .. x_P, y, 0 ..] this lane feeds as operand A of group q
#pragma unroll
    for (int t = 0; t < TN; t++) {
      const int sl = t >> 2, g = t & 3;
      const long long i = (long long)(64 * wave + cl_a + g) + (long long)NT * sl;
      double a = 0.0;
      if (i < A.n) {
        if (f < P) a = A.X[(long long)f * A.n + i];
        else if (f == P) a = A.y[i];
      }
      areg[q][t] = a;
    }
  }
  // result lane L: chain j = L % 4, canonical lane 64 w + 16 (L/16) + 4 ((L/4)%4) + g
  const int jch = lane & 3;
  const int cl_d = 16 * (lane >> 4) + 4 * ((lane >> 2) & 3);
  unsigned vbits = 0;    // validity of this lane's 4 results in the LAST slot (all earlier slots are full)
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int l = 64 * wave + cl_d + g;
    if ((long long)l + (long long)NT * (NS - 1 + (EXT ? A.mf_next : 0)) < A.n) vbits |= 1u << g;
  }
  if (tid < k) {
    s_par[0 * PIPE_KMAX + tid] = A.mu[tid];
    s_par[1 * PIPE_KMAX + tid] = A.scale[tid];
    s_par[2 * PIPE_KMAX + tid] = A.lb[tid];
    s_par[3 * PIPE_KMAX + tid] = A.ub[tid];
  }
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }

  // ---- owner state (waves 0..3), as in mh_sweep_spec
  const int myc = wave;
  const bool owner = (myc < ncw);
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + (owner ? myc : 0));
  const bool plane = owner && (lane < k);
  const int jl = (lane < k) ? lane : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  int zidx = 0;
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  if (zidx > kz - 1) zidx = kz > 0 ? kz - 1 : 0;   // lanes without a variate of their own read a valid neighbour (value unused)
  double th0 = plane ? A.theta0[(long long)cl * k + lane] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  // (a continuation window of a long call takes over what the windows before it left: SweepArgs.win_cont)
  // (accept counts are per launch and launch_sweep adds the windows up: taking the earlier count over in here cost 1.7 % per
  //  step -- the kernel sits at the SGPR limit and one more value live across the loop tipped the allocation, 11 -> 24 spills)
  int nacc = 0;
  int status = (A.win_cont && owner) ? A.status[cl] : FMCMC_CHAIN_OK, thin_ctr = A.thin_ctr0;
  unsigned int bitword = 0;
  // Row stores: until round 3 the byte offsets were 32-bit from the BUFFER bases, which capped a call at 4 GiB of samples
  // (1024 chains x 5 parameters: 1.04e5 kept rows) and sent anything longer to the general kernel.  Now the 64-bit part of an
  // address is the chain's own block and only the offset inside it is 32 bits (k ldS 8 < 2^32: the dispatcher's condition).
  // (wave-uniform 64-bit bases -- the chain's block -- plus a 32-bit byte offset per lane: one scalar-base store each)
  char* const s_base = reinterpret_cast<char*>(A.samples) + (BIG ? ((long long)cl * k) * A.ldS * 8 : 0ll);
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + (BIG ? ((long long)cl * k) * A.ldS * 8 : 0ll) : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (BIG ? (long long)cl * A.ldS * 8 : 0ll) : nullptr;
  const unsigned int lane_off = (unsigned int)(((BIG ? 0ll : (long long)cl * k) + jl) * A.ldS * 8);
  const unsigned int lp_off = BIG ? 0u : (unsigned int)((long long)cl * A.ldS * 8);
  unsigned int srow8 = 0;
  // (wave-uniform 64-bit base + a 32-bit offset per lane and row: nsteps kz 8 < 2^32 is the dispatcher's condition)
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + (BIG ? ((long long)cl * nsteps) * kz * 8 : 0ll);   // wave-uniform
  const unsigned int z_lane = (unsigned int)(((BIG ? 0ll : (long long)cl * nsteps) * kz + zidx) * 8);
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  // every lane of an owner wavefront keeps one variate and the log-uniform of the next step in flight (unconditional,
  // clamped addresses: a conditional load costs a register copy behind the load, i.e. an exposed wait)
  const double mu_l = A.mu[jl], sc_l = A.scale[jl];
  double z_nx = (owner && kz > 0) ? ld_z(nsteps >= 2 ? 1 : 0) : 0.0;
  double lu_nx = owner ? lu_row[nsteps >= 2 ? 1 : 0] : 0.0;
  auto logpost_of = [&](double tot, double sigma) -> double {
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) {
      f = fmh_nan();
    } else if (sigma == 0.0) {
      f = -fmh_inf();
    } else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) {   // (bits_stride: words per chain of the whole call's bitmap, set by launch_sweep for every launch)
      unsigned int* w = A.accept_bits + ((long long)cl * A.bits_stride + ((i - 1) >> 5));
      // the first word of a continuation window also holds the last bit of the window before it
      *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
    }
    bitword = 0;
  };
  lds_barrier();

  constexpr bool dbg = DBG;
  bool st_keep = false;                      // row of the step just decided, stored after the barrier
  bool st_acc = false;
  double st_th0 = 0.0, st_th1 = 0.0, st_f1 = 0.0;
  unsigned long long te = 0, tb1 = 0, to = 0, tb2 = 0, tf = 0, tc = 0, td = 0;
  for (int v = 1; v <= nsteps; v++) {
    unsigned long long t_0 = dbg ? clk() : 0;
    // ================= evaluation of version v of all 4 chains =================
    {
      const double* tj = s_th1 + jch * PIPE_KMAX;
      double bop[NG];                                            // B[k][blk][j] of group q
#pragma unroll
      for (int q = 0; q < NG; q++) {
        const int f = 4 * q + feat;
        bop[q] = (f < P) ? tj[ic + f] : (f == P ? -1.0 : 0.0);
      }
      const double cop = ic ? tj[0] : 0.0;                      // C = intercept of chain j
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      // (EXT) the first streamed slots are requested BEFORE the resident ones run: their L2 round trip hides under ~1200 cycles of MFMAs
      typedef double mf_d4 __attribute__((ext_vector_type(4)));
      constexpr int RD = (NG == 1) ? 4 : 2;                       // slots in flight per wave (register ring)
      const int next = EXT ? A.mf_next : 0;
      const mf_d4* sp = reinterpret_cast<const mf_d4*>(A.mf_stream) + ((long long)wave * next * NG) * 64 + lane;
      mf_d4 ring[EXT ? RD : 1][NG];
      if constexpr (EXT) {
        // (round 5: next == 0 -- up to 512 observations at 8 .. 15 covariates -- streams nothing, the resident slot is the last: these
        //  requests then read slot 0 of a stand-in buffer and nobody uses them; a branch around them cost the streamed forms 7 - 10 %)
        const int rlast = next > 0 ? next - 1 : 0;
#pragma unroll
        for (int r = 0; r < RD; r++)
#pragma unroll
          for (int q = 0; q < NG; q++) ring[r][q] = sp[(((r < next) ? r : rlast) * NG + q) * 64];
      }
      // batches of MB independent MFMA chains followed by their MB dependent FMAs: the result latency of one MFMA is
      // covered by issuing the next ones, and the batch shape (not the allocator's leftovers) bounds the live results
#pragma unroll
      for (int t0 = 0; t0 < TN; t0 += MB) {
        {
          constexpr int LAST = TN - 4;             // first pair of the last slot
          const int nu = (TN - t0 < MB) ? TN - t0 : MB;
          double d[MB];
          // Only the last slot has padding.  A padded observation has A = 0 in every group, so its result is the C
          // operand: feeding 0 instead of the intercept there makes -r == 0 exactly, and the accumulation below is the
          // same straight-line code for every batch (masking the RESULTS put selects in front of the last FMAs).
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) {
              const int t = t0 + u;
              const double cm = ((!EXT || next == 0) && t >= LAST) ? (((vbits >> (t - LAST)) & 1u) ? cop : 0.0) : cop;
              d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[0][t], bop[0], cm, 0, 0, 0);
            }
#pragma unroll
          for (int q = 1; q < NG; q++)
#pragma unroll
            for (int u = 0; u < MB; u++)
              if (u < nu) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[q][t0 + u], bop[q], d[u], 0, 0, 0);
          // d = -r of 16 observations x 4 chains per pair
#pragma unroll
          for (int u = 0; u < MB; u++)
            if (u < nu) acc[u & 3] = fmh_fma(d[u], d[u], acc[u & 3]);
        }
      }
      if constexpr (EXT) {
        // the streamed slots: operands of slot e, group q, for this lane = 4 doubles at ((wave next + e) NG + q) 64 + lane
        double cml[4];                                              // C operands of the LAST slot (padding: 0, see above)
#pragma unroll
        for (int g = 0; g < 4; g++) cml[g] = ((vbits >> g) & 1u) ? cop : 0.0;
        for (int e0 = 0; e0 < next; e0 += RD) {
#pragma unroll
          for (int r = 0; r < RD; r++) {
            const int e = e0 + r;
            if (e < next) {                                         // (uniform)
              mf_d4 a[NG];
#pragma unroll
              for (int q = 0; q < NG; q++) a[q] = ring[r][q];
              const int en = (e + RD < next) ? e + RD : next - 1;   // refill this ring position (clamped: the tail re-reads)
#pragma unroll
              for (int q = 0; q < NG; q++) ring[r][q] = sp[(en * NG + q) * 64];
              const bool last = (e == next - 1);
              double d[4];
#pragma unroll
              for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0][g], bop[0], last ? cml[g] : cop, 0, 0, 0);
#pragma unroll
              for (int q = 1; q < NG; q++)
#pragma unroll
                for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[q][g], bop[q], d[g], 0, 0, 0);
#pragma unroll
              for (int g = 0; g < 4; g++) acc[g] = fmh_fma(d[g], d[g], acc[g]);
            }
          }
        }
      }
      // canonical levels 1..32 inside the wave (see the mapping above); a lane only ever adds canonical partial sums, and
      // the sums that matter end up in the lanes of block 3
      double fs = (acc[0] + acc[1]) + (acc[2] + acc[3]);        // levels 1, 2: the four accumulators
      fs = fs + dpp_d<0x114>(fs);                               // level 4: row_shr:4, blocks 1 and 3 hold (b, b - 1)
      fs = fs + dpp_d<0x118>(fs);                               // level 8: row_shr:8, block 3 holds all four
      {
        const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
        const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
             __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));      // level 16: rows r, r ^ 1
      }
      {
        const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
        const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
             __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));      // level 32: rows r, r ^ 2
      }
      if (lane >= 60) s_fold[jch * NW + wave] = fs;             // lanes 60..63: block 3 of row 3, chains 0..3
    }
    // (Measured: moving log(sigma) of the owners in front of, or right behind, their MFMAs makes the step SLOWER:
    //  fp64 VALU work issued while the SIMD partner runs MFMAs slows those -- one fp64 datapath -- whereas in
    //  the owner phase below that datapath is idle.)
    // sigma-only part of the closed form: the owner waves are the first of their SIMD to finish their MFMAs (the older
    // wave wins the arbitration) and would wait ~1200 ticks at the barrier; its ~65 fp64 instructions run there, in the
    // shadow of the partner wave's MFMAs, instead of in the exposed owner phase.  The results are wave-uniform (SGPRs).
    double sigma = 0.0, nt1_fast = 0.0, ss_fast = 1.0, rs_fast = 1.0;
    bool sg_fast = false, ok_fast = false;
    if (owner) {
      sigma = readlane_d(th1, k - 1);
      const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
      sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;                     // positive, finite, normal
      const double sg = sg_fast ? sigma : 1.0;
#ifdef MFO_NOLOG   /* timing ablation (tools/exp_mfmar.hip): what the logarithm in the owners' MFMA slack costs */
      const double t1_fast = sg + FMH_K(FMH_LN_SQRT_2PI);
#else
      const double t1_fast = fmh_log_pn(sg) + FMH_K(FMH_LN_SQRT_2PI);   // same bits as fmh_log(sigma) on this range
#endif
      nt1_fast = uniform_d(dn * t1_fast);
      ss_fast = uniform_d(sg * sg);
      // denominator half of (0.5 tot) / sigma^2 (mh_common.hpp: div_recip / div_finish), in the same slack
      ok_fast = sg_fast && mfr_div_safe(ss_fast);
      rs_fast = uniform_d(div_recip(ok_fast ? ss_fast : 1.0));
    }
    unsigned long long t_1 = dbg ? clk() : 0;
    lds_barrier();
    unsigned long long t_2 = dbg ? clk() : 0;
    // ================= owners: fold, decide, propose =================
    if (owner) {
      __builtin_amdgcn_s_setprio(3);
      double wsum = s_fold[myc * NW + (lane & 7)];              // the wave sums of this chain, one per lane of an octet
      // increment of the NEXT proposal: consumes the variate fetched behind the previous step's releasing barrier, a whole
      // evaluation ago.  The only vector-memory wait of the phase sits HERE, on those loads (and on the row stores issued
      // with them); the registers are refilled behind the barrier below.
      double lu = lu_nx, zc = z_nx;
      asm volatile("" : "+v"(lu), "+v"(zc));
      const double dz = mu_l + sc_l * zc;   // (unused by fixed / idle lanes; mu and scale live in registers: no LDS read, no exec region)
      wsum = wsum + dpp_d<0xB1>(wsum);                          // level 64:  waves w, w ^ 1 (quad_perm [1,0,3,2])
      wsum = wsum + dpp_d<0x4E>(wsum);                          // level 128: quad_perm [2,3,0,1]
      const double tot = wsum + dpp_d<0x141>(wsum);             // level 256: row_half_mirror (quads are uniform)
      unsigned long long t_a = dbg ? clk() : 0;
      // The owner phase is exposed, and a wave issues one instruction per ~6.5 cycles whatever their dependences: the
      // common case is straight-line (three instructions finish the division, one compare, selects, no guard: -inf needs
      // none and a NaN ends in the ratio), everything else sits behind ONE wave-level branch and redoes the closed form
      // with the general code.
      const double h = 0.5 * tot;
      double f1 = -nt1_fast - div_finish(h, ss_fast, rs_fast);
      unsigned long long t_b = dbg ? clk() : 0;
      const double th1_eval = th1;
      bool keep_row = true, acc = false;
      const double ratio_f = f1 - f0;
      const bool rare = (v == 1) || (status != FMCMC_CHAIN_OK) || !ok_fast || !mfr_div_safe(h) || fmh_isnan(ratio_f);
      auto propose = [&](bool frozen) {   // next proposal (a failed chain keeps its theta1; behind the last step it goes nowhere)
        double t = th0 + dz;
        if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) {
          if (plane && !fixed_l) t = reflect1(t, s_par[2 * PIPE_KMAX + lane], s_par[3 * PIPE_KMAX + lane]);
        }
        t = fixed_l ? th0 : t;
        th1 = frozen ? th1 : t;
        if (plane) s_th1[myc * PIPE_KMAX + lane] = th1;
      };
      double th0_row;
      if (__builtin_expect(!__any(rare), 1)) {
        acc = lu < ratio_f;
        th0 = acc ? th1 : th0;
        f0 = acc ? f1 : f0;
        th0_row = th0;
        propose(false);
      } else {
        keep_row = false;
        if (sg_fast) {
          f1 = -nt1_fast - h / ss_fast;
          if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
        } else {
          f1 = logpost_of(tot, sigma);
        }
        if (v == 1) {
          f0 = f1;
          keep_row = true;
        } else if (status == FMCMC_CHAIN_OK) {
          const double ratio = f1 - f0;
          if (fmh_isnan(f1) || fmh_isnan(ratio)) {
            status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
            if (lane == 0) { A.status[cl] = status; A.status_step[cl] = v + A.step_off; }
            if (plane) A.status_theta[(long long)cl * k + lane] = th1;
            flush_bits(v);
          } else {
            acc = lu < ratio;
            keep_row = true;
          }
        }
        th0 = acc ? th1 : th0;
        f0 = acc ? f1 : f0;
        th0_row = th0;
        propose(status != FMCMC_CHAIN_OK);
      }
      st_acc = acc;
      unsigned long long t_c = dbg ? clk() : 0;
      if (dbg) { tf += t_a - t_2; tc += t_b - t_a; td += t_c - t_b; }
      __builtin_amdgcn_s_setprio(0);
      st_keep = keep_row; st_th0 = th0_row; st_th1 = th1_eval; st_f1 = f1;
    }
    unsigned long long t_3 = dbg ? clk() : 0;
    lds_barrier();
    if (dbg) { unsigned long long t_4 = clk(); te += t_1 - t_0; tb1 += t_2 - t_1; to += t_3 - t_2; tb2 += t_4 - t_3; }
    // Row stores and bookkeeping of step v happen AFTER the barrier that releases the next evaluation: the owner
    // waves are the first of their SIMD to finish their MFMAs (~1400 ticks of slack), the stores ride in that slack
    // instead of sitting in the exposed owner phase.
    if (owner) {
      // The refills of the two stream registers (consumed in the owner phase of the NEXT step, a whole evaluation away) and the
      // accept bookkeeping ride here too (round 3: -1.9 % per step; they were 13 instructions of the exposed owner phase).
      lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
      if (kz > 0) z_nx = ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1);
      nacc += st_acc ? 1 : 0;
      bitword |= (st_acc ? 1u : 0u) << ((v - 1) & 31);
      if (st_keep && v > burnin) {
        thin_ctr += 1;
        if (thin_ctr == thin) {
          thin_ctr = 0;
          if (plane) {
            *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = st_th0;
            if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = st_th1;
          }
          if (l_base && lane == 0 && !dbg) *reinterpret_cast<double*>(l_base + (lp_off + srow8)) = st_f1;
          srow8 += 8;
        }
      }
      if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    }
  }
  if (dbg && lane == 0 && A.logpost) {   // stamps leave through the logpost buffer in this diagnostic mode
    double* d = A.logpost + (long long)A.nchains * A.S - 8 * ((long long)blockIdx.x * NW + wave + 1);
    d[0] = (double)te; d[1] = (double)tb1; d[2] = (double)to; d[3] = (double)tb2; d[4] = (double)nsteps; d[5] = (double)tf; d[6] = (double)tc; d[7] = (double)td;
  }
  if (owner) {
    if (plane) A.theta0[(long long)cl * k + lane] = th0;
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;   // (of THIS launch: launch_sweep adds the windows up)
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    }
  }
}

size_t mfma_lds_bytes() { return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 * NW); }

}  // namespace
