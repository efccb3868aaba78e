// mh_wide2.hpp -- mh_sweep_wide2<KIND>: the observation-sharded sweep of wide Gaussian linear models (config C4: kernel_ram,
// k = 50, 512 chains per GPU) as a DATAFLOW kernel.
//
// Same decomposition as the sharded instantiations of mh_sweep_kernel (mh_common.hpp, eval_sharded): workgroup b of 256
// owns the canonical lanes 2b, 2b + 1 of ALL chains -- a constant slice of <= 40 observations, kept in LDS in fp64-MFMA
// operand layout -- and the chains 2b, 2b + 1; per step the proposals of all chains cross the chip one way and the lane
// partials the other way.  What is different is WHEN things happen.  There a step was one sequence for all 512 threads:
// propose | hand-over | matrix cores | hand-over | adapt, accept -- 13 us of one-wave scalar work with the matrix cores
// idle, 14 us of matrix-core work with the owners idle, 10 us of hand-over latency with everything idle.  Here
//   * the chains form two groups (even / odd chain of every workgroup) that run HALF A STEP OUT OF PHASE;
//   * waves 0 and 1 are the OWNERS of the workgroup's two chains and never evaluate: wait for the partials of their chain,
//     finish the log-posterior, adapt the factor (kernel_ram), accept, store, propose, publish -- and draw the variates of
//     the step after next while they wait;
//   * waves 2..7 are EVALUATORS and never decide anything: wait for a group's proposals, run that group's slice product
//     on the matrix cores (shard_columns_mfma, chain set {2l + g}), publish the lane partials, turn to the other group;
//   * the two kinds of waves meet only through the exchange tables in HBM and four grid-wide arrival counters (per group:
//     proposals complete, partials complete), each waited for by ONE lane per workgroup.
// So one group's scalar phase and hand-over latencies run under the other group's matrix-core work.  Arithmetic, its order
// and the canonical tree are those of every other kernel: bit-identical results.
// (Knob groups=4: four groups a quarter step apart, chain c in group c % 4.  Measured at C4: 30.5 us per step against 25.7
//  with two -- a visit of the evaluators costs 2.3 us + 0.8 us per N-tile, so halving the tiles per visit does not halve it,
//  and eight counter sets in flight lengthen every hand-over.  Kept as a tested alternative, not the default.)
//
// Hand-over protocol (MI355X_MICROARCH.md, inter-workgroup visibility; the form the sharded kernels have used since round
// 1): payload moved with agent-scope (sc1) stores and loads only; every storing wave drains its stores (s_waitcnt vmcnt(0))
// before it, or the last of its workgroup's storing waves (told by a counter in LDS), adds to the arrival counter; a
// consumer loads only after the ONE polling lane of its workgroup has seen the counter complete (other waves: after an LDS
// word that wave then sets).  Every spin is bounded; a hand-over that does not complete marks the chains of the workgroup
// FMCMC_CHAIN_SYNC_TIMEOUT and lets every loop run out.
#pragma once

namespace {

constexpr int W2_NEVAL = NW - 2;       // evaluator waves
constexpr int W2_BARW = 32 * 20;       // 32-bit words of one arrival counter set: arrive[8] | (unused) | top replicas[8], one 128-byte line each

// Grid-wide "everybody has published" in two levels: workgroup b adds (returning) to shard b % 8 (one XCD under round-robin
// placement); the last of a shard adds -- not waited for -- to each of EIGHT REPLICAS of the top counter; ONE wave per
// waiting workgroup polls the replica of its own shard until it holds (shards that take part) x epoch.  Eight adds and 32
// pollers per replica line and epoch, no release stage (the sequential form, shard_barrier, has one: last of the shards ->
// release words; here that was 0.2 us per step slower).  Tried and dropped, both bit-identical: every waiter polling the
// eight (sharded, non-returning) arrival counters directly: 27.5 -> 40.9 us per step, 512 pollers and 256 atomic adds fight
// for the same eight lines; per-producer flag words polled by every consumer wave (no atomics at all): 38.8 us, 2048 polling
// waves on the fabric.  Shards that take part: 8; 4 for the proposals of one of FOUR chain groups (knob groups=4), where only
// the workgroups of one parity, i.e. every second shard, own chains of the group.
__device__ __forceinline__ void w2_arrive(unsigned* bar, unsigned epoch) {   // ONE lane, after the drain of every wave it signals for
  const unsigned ngroups = 8, gsize = gridDim.x / 8, g = blockIdx.x % ngroups;
  unsigned* rep = bar + 9 * 32;
  const unsigned old = __hip_atomic_fetch_add(&bar[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1 == epoch * gsize) {   // the last of its shard
    for (unsigned q = 0; q < ngroups; q++) (void)__hip_atomic_fetch_add(&rep[q * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
typedef __attribute__((address_space(3))) unsigned* w2_ldsu_t;
__device__ __forceinline__ unsigned w2_lds_ld(unsigned* p) {   // (LDS address space: a generic pointer would make these FLAT accesses)
  return __hip_atomic_load((w2_ldsu_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void w2_lds_st(unsigned* p, unsigned v) {
  __hip_atomic_store((w2_ldsu_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// lane 0 of the calling wave polls its shard's replica of the top counter for `target` = shards that take part x epoch;
// wave-uniform result; false: timed out (or the workgroup is already lost)
__device__ __forceinline__ bool w2_wait(unsigned* bar, unsigned target, unsigned* s_lost, unsigned limit = 20000000u) {
  const unsigned g = blockIdx.x % 8;
  const unsigned epoch = target;
  unsigned* rel = bar + 9 * 32;
  bool ok = true;
  if ((threadIdx.x & 63) == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(&rel[g * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > limit || w2_lds_ld(s_lost)) { ok = false; break; }
    }
  }
  // Acquire side of the hand-over FOR THE COMPILER: the payload is read with relaxed agent-scope (sc1) loads, which the
  // hardware serves from the coherent level once the counter has been seen (MI355X_MICROARCH.md, hand-offs with sc1 loads in
  // place of the acquire; an agent-scope acquire fence would be a buffer_inv sc1 that also drops this workgroup's slice from
  // the caches: 11.8 -> 32.6 us per step, tools/scalar_slice_probe.hip).  What relaxed atomics do NOT give is an order the
  // optimiser has to respect -- it may hoist a payload load above the polling loop -- so the wait ends in a compiler-level
  // barrier (no instruction).  The producer side has one in the s_waitcnt vmcnt(0) asm in front of every w2_arrive.
  asm volatile("" ::: "memory");
  return __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
}

// kernel_ram on an owner wave of this kernel: lane = row of the lower factor, as in mh_streamed.hpp (ram_propose_rows,
// ram_update_rows, ram_update_propose_rows: same operations on every element in the same order, same bits), but S and G
// INTERLEAVED -- SG[(i LD + j)] = (S_ij, G_ij) -- and the coefficients of the update as pairs C[j] = (d_j, kappa_j): one
// two-address LDS instruction per pair instead of one per double, and no predicate per column: above the diagonal S and G
// hold +0 and stay +0 (fma(+0, kappa, +0 d) = +0 for the finite kappa and d > 0 that ram_coef lets through; the proposal's
// chain adds nw z = +-0 there, as it always has).  21 -> 9 instructions per column of the fused pass, which at k = 50 was
// 5.2 of the owner's 8 us between the partials and the next proposal.
// (software pipeline: two register sets used alternately, the loads of a group of four columns issued a whole group ahead
//  of its arithmetic, every address an immediate offset from three pointers that move by one group -- copying a "next" set
//  into a "current" one made the compiler wait for the loads it had just issued)
// (the address as a value the optimiser knows nothing about: it then keeps `pointer + constant` as the instruction's
//  unsigned offset field instead of re-deriving every address from the loop's start with an add of its own)
#define W2_OPAQUE(ptr) { unsigned a_ = (unsigned)(unsigned long long)(ptr); asm volatile("" : "+v"(a_)); (ptr) = (lds_dptr_t)(unsigned long long)a_; }
#define W2_LOAD4(S_, G_, D_, K_, Z_, rp, cp, zp)                                                         \
  _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                        \
    S_[u] = (rp)[2 * u]; G_[u] = (rp)[2 * u + 1]; D_[u] = (cp)[2 * u]; K_[u] = (cp)[2 * u + 1]; Z_[u] = (zp)[u]; }
#define W2_STEP4(S_, G_, D_, K_, Z_, rp)                                                                 \
  _Pragma("unroll") for (int u = 3; u >= 0; u--) {                                                       \
    const double nw = fmh_fma(G_[u], K_[u], S_[u] * D_[u]);                                              \
    (rp)[2 * u] = nw; (rp)[2 * u + 1] = s;                                                               \
    s = fmh_fma(nw, Z_[u], s); }
__device__ __attribute__((noinline)) double w2_ram_update_propose(lds_dptr_t SG, lds_dptr_t C, lds_dptr_t z, int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  double s = 0.0;
  if (lane < kf) {
    const lds_dptr_t row = SG + 2 * lane * LD;
    const int ng = kf >> 2;                    // full groups of four columns: columns 0 .. 4 ng - 1
    for (int j = kf - 1; j >= 4 * ng; j--) {   // the (at most three) columns above them come first
      const double nw = fmh_fma(row[2 * j + 1], C[2 * j + 1], row[2 * j] * C[2 * j]);
      row[2 * j] = nw;
      row[2 * j + 1] = s;
      s = fmh_fma(nw, z[j], s);
    }
    if (ng > 0) {
      int g = ng - 1;                          // group g: columns 4 g .. 4 g + 3
      lds_dptr_t rq = row + 8 * g, cq = C + 8 * g, zq = z + 4 * g;
      double s0[4], g0[4], d0[4], k0[4], z0[4], s1[4], g1[4], d1[4], k1[4], z1[4];
      W2_LOAD4(s0, g0, d0, k0, z0, rq, cq, zq)
      while (g >= 2) {                         // (pointers at the LOWEST of the three groups in flight: LDS offsets are unsigned)
        rq -= 16; cq -= 16; zq -= 8;
        W2_OPAQUE(rq) W2_OPAQUE(cq) W2_OPAQUE(zq)
        W2_LOAD4(s1, g1, d1, k1, z1, rq + 8, cq + 8, zq + 4)
        W2_STEP4(s0, g0, d0, k0, z0, rq + 16)
        W2_LOAD4(s0, g0, d0, k0, z0, rq, cq, zq)
        W2_STEP4(s1, g1, d1, k1, z1, rq + 8)
        g -= 2;
      }
      if (g == 1) {
        rq -= 8; cq -= 8; zq -= 4;
        W2_OPAQUE(rq) W2_OPAQUE(cq) W2_OPAQUE(zq)
        W2_LOAD4(s1, g1, d1, k1, z1, rq, cq, zq)
        W2_STEP4(s0, g0, d0, k0, z0, rq + 8)
        W2_STEP4(s1, g1, d1, k1, z1, rq)
      } else {
        W2_STEP4(s0, g0, d0, k0, z0, rq)
      }
    }
  }
  return s;
}
__device__ __attribute__((noinline)) double w2_ram_propose(lds_dptr_t SG, lds_dptr_t z, int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  double s = 0.0;
  if (lane < kf) {
    const lds_dptr_t row = SG + 2 * lane * LD;
    int j = kf - 1;
    double sc[4], zc[4];
    if (j >= 3) {
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = row[2 * (j - u)]; zc[u] = z[j - u]; }
    }
    for (; j >= 3; j -= 4) {
      const int jn = (j - 4 >= 3) ? j - 4 : j;
      double sn[4], zn[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { sn[u] = row[2 * (jn - u)]; zn[u] = z[jn - u]; }
      const double g0 = s, g1 = fmh_fma(sc[0], zc[0], g0), g2 = fmh_fma(sc[1], zc[1], g1), g3 = fmh_fma(sc[2], zc[2], g2);
      s = fmh_fma(sc[3], zc[3], g3);
      row[2 * j + 1] = g0; row[2 * (j - 1) + 1] = g1; row[2 * (j - 2) + 1] = g2; row[2 * (j - 3) + 1] = g3;
#pragma unroll
      for (int u = 0; u < 4; u++) { sc[u] = sn[u]; zc[u] = zn[u]; }
    }
    for (; j >= 0; j--) {
      const double sij = row[2 * j];
      row[2 * j + 1] = s;
      s = fmh_fma(sij, z[j], s);
    }
  }
  return s;
}
__device__ __attribute__((noinline)) void w2_ram_update(lds_dptr_t SG, lds_dptr_t C, int LD_, int kf_) {
  const int lane = threadIdx.x & 63;
  const int LD = __builtin_amdgcn_readfirstlane(LD_), kf = __builtin_amdgcn_readfirstlane(kf_);
  if (lane < kf) {
    const lds_dptr_t row = SG + 2 * lane * LD;
    int j = 0;
    for (; j + 3 < kf; j += 4) {
      double nw[4];
#pragma unroll
      for (int u = 0; u < 4; u++) nw[u] = fmh_fma(row[2 * (j + u) + 1], C[2 * (j + u) + 1], row[2 * (j + u)] * C[2 * (j + u)]);
#pragma unroll
      for (int u = 0; u < 4; u++) row[2 * (j + u)] = nw[u];
    }
    for (; j < kf; j++) row[2 * j] = fmh_fma(row[2 * j + 1], C[2 * j + 1], row[2 * j] * C[2 * j]);
  }
}

__host__ __device__ inline size_t wide2_lds_doubles(int k, int kf, int kind, int kz, int mblk) {
  return 4 * (size_t)k + (k / 2 + 1) + 4 * (size_t)(kz + 1) + 6 + 2 * (size_t)chain_lds_doubles(k, kf, kind) + 2 + (size_t)mblk;
}

template <int KIND, int NMT>
__global__ __launch_bounds__(NT) void mh_sweep_wide2(const SweepArgs A0) {
  SweepArgs A = A0;
  A.kind = KIND;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz, p = A.p, ic = A.intercept, nb = ic + p;
  const int NC = (int)A.nchains, NCP = NC + SH_PAD;
  const int NH = (NC + A.sh_ngrp - 1) / A.sh_ngrp;   // the proposal table keeps each group's chains together: row j = [group 0 | group 1 | ..]
  // ---- LDS: kernel parameters | which[] | variates [2 chains][2 parities][kz + 1] | sync words | 2 chain blocks | slice block
  double* s_mu = smem;
  double* s_scale = s_mu + k;
  double* s_lb = s_scale + k;
  double* s_ub = s_lb + k;
  int* s_which = (int*)(s_ub + k);
  double* s_z = s_ub + k + (k / 2 + 1);
  unsigned* s_sync = (unsigned*)(s_z + 4 * (kz + 1));     // [0..3] proposals of group q seen (epoch), [4..7] evaluator arrivals, [8] lost, [9] kf
  double* s_chains = s_z + 4 * (kz + 1) + 6;
  const int ng = A.sh_ngrp;                                // chain groups: 2 (chain c in group c % 2) or 4 (c % 4)
  if (tid == 0) {
    int kf0 = 0;
    for (int j = 0; j < k; j++)
      if (!A.fixed[j]) s_which[kf0++] = j;
    for (int q = 0; q < 9; q++) s_sync[q] = 0;
    s_sync[9] = (unsigned)kf0;
  }
  if (tid < k) {
    s_mu[tid] = A.mu[tid];
    s_scale[tid] = A.scale[tid];
    s_lb[tid] = A.lb[tid];
    s_ub[tid] = A.ub[tid];
  }
  __syncthreads();
  const int kf = (int)s_sync[9];
  const int LD = kf | 1;
  const int CHS = chain_lds_doubles(k, kf, KIND);
  double* s_blk = s_chains + 2 * CHS + ((2 * CHS) & 1);
  {
    const double* src = A.sh_mfma + (long long)blockIdx.x * A.sh_mblk;
    for (int i = tid; i < A.sh_mblk; i += NT) s_blk[i] = src[i];
  }
  __syncthreads();
  const int nsteps = (int)A.nsteps;
  unsigned* const s_lost = &s_sync[8];

  if (wave >= 2) {
    // =========================================== evaluator waves ===========================================
    // N-tiles of a group dealt so that the four SIMDs get the same number (waves w and w + 4 share SIMD w % 4; waves 0, 1 are
    // the owners): SIMD 0 -> wave 4, SIMD 1 -> wave 5, SIMD 2 -> waves 2, 6 alternating, SIMD 3 -> waves 3, 7
    // The single evaluator waves of SIMDs 0 and 1 share their SIMD with an owner wave and have no partner to overlap a tile's
    // epilogue with: of 16 tiles they take 3 each (wave 4: 0, 1, 2; wave 5: 3, 4, 5), the pairs of SIMDs 2 and 3 take 5
    // (waves 2 / 6: 6, 8, 10 / 7, 9; waves 3 / 7: 11, 13, 15 / 12, 14).  Knob tiles=0: the even split 4 | 4 | 2 + 2 | 2 + 2.
    // (measured and dropped, round 3, after the T10 form: the single evaluator of an owner's SIMD working only in the visits of that
    //  owner's OWN group -- 4 or 5 tiles, the pairs 12 or 11 -- and sitting out the other group's, which is when the owner runs its
    //  critical section, slowed 1.66x by an evaluator next to it (tools/exp_shard_mfma.hip, mode 3): 17.2 / 17.3 us per step against
    //  16.9 -- what the owner gains, the longer visits of the pairs lose)
    int tfirst, tstep, tcount = 0;
    if (A.sh_tiles == 0) {
      tfirst = (wave == 4) ? 0 : (wave == 5) ? 1 : (wave == 2) ? 2 : (wave == 6) ? 6 : (wave == 3) ? 3 : 7;
      tstep = (wave == 4 || wave == 5) ? 4 : 8;
    } else {
      tfirst = (wave == 4) ? 0 : (wave == 5) ? 3 : (wave == 2) ? 6 : (wave == 6) ? 7 : (wave == 3) ? 11 : 12;
      tstep = (wave == 4 || wave == 5) ? 1 : 2;
      tcount = (wave == 6 || wave == 7) ? 2 : 3;
    }
    ShardMfma sm;
    sm.th = A.sh_th; sm.part = A.sh_part; sm.p = p; sm.ic = ic; sm.lane0 = (int)blockIdx.x * 2; sm.tcount = tcount;
    sm.lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const double*)s_blk;
    sm.ncp = NCP; sm.cstride = ng; sm.tfirst = tfirst; sm.tstep = tstep;
    const bool t10 = __builtin_amdgcn_readfirstlane(A.sh_t10) != 0;
    bool lost = false;
#ifdef FMCMC_STAMP
#define W2_EVENT(cond, idx) do { if ((cond) && lane == 0 && k >= 48 && (long long)blockIdx.x * 2 < A.nchains) \
    A.status_theta[(long long)blockIdx.x * 2 * k + (idx)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
    unsigned long long ev_acc[4] = {0, 0, 0, 0}, ev_prev = stamp_clk();
#define W2_EV_STAMP(i) do { if (wave == 2 || FMCMC_STAMP_WAVE < 0) { const unsigned long long t_ = stamp_clk(); ev_acc[i] += t_ - ev_prev; ev_prev = t_; } } while (0)
#else
#define W2_EV_STAMP(i) do { } while (0)
#define W2_EVENT(cond, idx) do { } while (0)
#endif
    for (int v = 1; v <= nsteps && !lost; v++) {
      for (int g = 0; g < ng && !lost; g++) {
        unsigned* X1 = A.sh_bar + (2 * g) * W2_BARW;
        unsigned* X2 = A.sh_bar + (2 * g + 1) * W2_BARW;
        // the proposals of group g, version v, are complete: wave 2 polls, the other evaluators watch an LDS word it then sets
        if (wave == 2) {
          const bool ok = w2_wait(X1, (unsigned)v * ((ng == 4) ? 4u : 8u), s_lost, (A.debug & 512) ? 400000u : 20000000u);   // (shards that own chains of the group)
          if (!ok) { if (lane == 0) w2_lds_st(s_lost, 1u); lost = true; }
          else if (lane == 0) w2_lds_st(&s_sync[g], (unsigned)v);
        } else {
          unsigned spins = 0;
          while (w2_lds_ld(&s_sync[g]) < (unsigned)v) {
            __builtin_amdgcn_s_sleep(1);
            if (w2_lds_ld(s_lost) || ++spins > ((A.debug & 512) ? 800000u : 40000000u)) { lost = true; break; }
          }
          asm volatile("" ::: "memory");   // (compiler-level acquire of the LDS relay, as in w2_wait)
        }
        if (lost) break;
        W2_EV_STAMP(0);
        W2_EVENT(wave == 2 && v == 300, 26 + 2 * g);          // proposals of group g, version 300, seen
        const int Ng = (NC + ng - 1 - g) / ng;                  // chains of the group in this launch
        // (measured and dropped, round 3: the second evaluator of SIMDs 2 / 3 starting a visit 0.2 .. 1.3 us late, so that the pair does
        //  not run its B loads, MFMA blocks and epilogues in lockstep -- MI355X_MICROARCH.md, two waves per SIMD, item 9 --: 18.6 ..
        //  18.8 us per step against 18.5; the pairs drift apart by themselves)
        if (Ng > 0) {
          sm.NC = Ng; sm.coff = g; sm.thoff = g * NH;
          shard_mfma_dispatch<2, NMT>(sm, (p + 3) >> 2, t10);    // (compile-time K-block counts; C4's width with its 10 values per lane group)
        }
        W2_EV_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's partials have been acknowledged
        W2_EV_STAMP(2);
        if (lane == 0) {
          const unsigned old = __hip_atomic_fetch_add((w2_ldsu_t)&s_sync[4 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old + 1 == (unsigned)W2_NEVAL * (unsigned)v) w2_arrive(X2, (unsigned)v);   // the last evaluator of the workgroup signals for all
        }
        W2_EV_STAMP(3);
        W2_EVENT(wave == 2 && v == 300, 27 + 2 * g);          // this wave is past its part of the arrival for the partials
        W2_EVENT(wave == 5 && v == 300, 36 + g);              // (another evaluator wave, for the spread inside a workgroup)
      }
    }
#ifdef FMCMC_STAMP
    if (wave == 2 && lane < 4 && k >= 32 && (long long)blockIdx.x * 2 < A.nchains)
      A.status_theta[(long long)blockIdx.x * 2 * k + 16 + lane] = (double)ev_acc[lane];
    // (-DFMCMC_STAMP_WAVE=-1: every evaluator wave, four words each behind wave 2's)
    if (FMCMC_STAMP_WAVE < 0 && wave > 2 && lane < 4 && k >= 44 && (long long)blockIdx.x * 2 < A.nchains)
      A.status_theta[(long long)blockIdx.x * 2 * k + 16 + 4 * (wave - 2) + lane] = (double)ev_acc[lane];
#endif
    return;
  }

  // =============================================== owner waves ===============================================
  const int slot = wave;                                       // chain slot of the workgroup
  const long long cl = (long long)blockIdx.x * 2 + slot;       // local chain
  const int g = (int)(cl % ng);                                // its group
  const bool has = cl < A.nchains;
  const unsigned int cgid = (unsigned int)(A.chain_base + cl);
  unsigned* X1 = A.sh_bar + (2 * g) * W2_BARW;
  unsigned* X2 = A.sh_bar + (2 * g + 1) * W2_BARW;
  ChainLds L = chain_lds(s_chains + slot * CHS, k, kf, KIND);
  double* const SG = L.SigA;      // kernel_ram: the pairs (S_ij, G_ij), [kf][LD][2] over SigA | SigB
  double* const CF = L.vz;        // kernel_ram: the pairs (d_j, kappa_j) of the pending update, [kf][2] over vz | vv
  double f0 = 0.0, f1 = 0.0;
  long long abs_iter = 0, nacc = 0;
  int nerr = 0, status = FMCMC_CHAIN_OK;
  unsigned int bitword = 0;
  if (has) {
    if (lane < k) {
      const double t = A.theta0[cl * k + lane];
      L.th0[lane] = t;
      L.th1[lane] = t;
    }
    if (KIND == FMCMC_KERNEL_RAM) {
      for (int e = lane; e < kf * LD; e += 64) {
        const int a = e / LD, b = e % LD;
        SG[2 * e] = A.fresh ? ((a == b) ? 1.0 * A.eps : 0.0) : ((b < kf && b <= a) ? A.Sigma[(cl * kf + a) * kf + b] : 0.0);
        SG[2 * e + 1] = 0.0;
      }
      if (!A.fresh) {
        abs_iter = A.abs_iter[cl];
        if (A.nerrors) nerr = A.nerrors[cl];
      }
    }
  }
  wave_sync_lds();
  // variates of loop step ii into the buffer of its parity: lanes 0..kz-1 the proposal's, lane kz the log accept uniform
  auto draw = [&](int ii) {
    if (!has || ii > nsteps || lane > kz) return;
    double* zb = s_z + (slot * 2 + (ii & 1)) * (kz + 1);
    const unsigned int st = (unsigned int)(A.step_base + ii);
    double v;
    if (lane == kz) {
      v = (A.rng_mode == FMCMC_RNG_FED) ? A.fed_logu[cl * A.nsteps + (ii - 1)] : fmh_log_accept_u(A.seed, st, cgid);
    } else {
      if (A.rng_mode == FMCMC_RNG_FED) v = A.fed_z[(cl * A.nsteps + (ii - 1)) * kz + lane];
      else if (KIND == FMCMC_KERNEL_RAM && A.ram_df > 0.0) v = fmh_student_t(A.seed, st, cgid, (unsigned int)lane, A.ram_df);
      else if (A.variate == 1) v = fmh_unif(A.seed, st, cgid, (unsigned int)lane);
      else v = fmh_normal(A.seed, st, cgid, (unsigned int)lane);
    }
    zb[lane] = v;
  };
  // publish theta1 of this chain ([coefficient][chain] table) and signal; a wave without a chain only signals
  auto publish = [&](unsigned epoch, bool store) {
    if (has && store && lane < nb) sh_store(&A.sh_th[(long long)lane * NCP + g * NH + (int)(cl / ng)], L.th1[lane]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (mode=512, tests: workgroup 1 never announces version 3 of its group-0 proposals -- a lost hand-over, exercised)
    if (lane == 0 && !((A.debug & 512) && blockIdx.x == 1 && epoch == 3 && g == 0)) w2_arrive(X1, epoch);
  };
  // row bookkeeping (R/mcmc.R:786-813), as in mh_sweep_kernel
  const int burnin = (int)A.burnin, thin = (int)A.thin;
  int thin_ctr = 0;
  long long srow = 0;
  double* const out_s = A.samples + (cl * k + (lane < k ? lane : 0)) * A.ldS;
  double* const out_d = A.draws ? A.draws + (cl * k + (lane < k ? lane : 0)) * A.ldS : nullptr;
  double* const out_l = A.logpost ? A.logpost + cl * A.ldS : nullptr;
  // What the decision of step i needs and f(theta1) does not enter, prepared while the wave waits for the partials:
  // the sigma-only part of the closed form of the pending proposal; for kernel_ram eta(i, k) = min(1, k i^(-2/3))
  // (R/kernel_ram.R:67) and the prefix sums of z^2 (their last one is |z|^2).
  // (measured and dropped, round 3: the reciprocal halves of (tot / 2) / sigma^2 and of the division by |z|^2 prepared here as
  //  well (div_recip / div_finish) and no exp when f1 - f0 >= 0 -- ~27 dependent instructions less in the owner's critical
  //  section: 16.4-16.5 us per step against 16.3-16.5, old and new library alternating on one box)
  double pre_nt1 = 0.0, pre_ss = 1.0, pre_eta = 0.0, pre_zl = 0.0, pre_Pj = 0.0, pre_Pj1 = 0.0, pre_nrm2 = 1.0;
  bool pre_sigma_ok = false;
  const double dn = (double)A.n;
  auto prepare = [&](int i) {
    if (!has) return;
    const double sigma = L.th1[k - 1];
    const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
    pre_sigma_ok = (sg_hi - 0x00100000u) < 0x7fe00000u;                 // positive, finite, normal: the closed form's main branch
    const double sg = pre_sigma_ok ? sigma : 1.0;
    pre_nt1 = dn * (fmh_log(sg) + FMH_LN_SQRT_2PI);
    pre_ss = sg * sg;
    if (KIND == FMCMC_KERNEL_RAM && i >= 2) {
      double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)i));
      if (eta > 1.0) eta = 1.0;
      pre_eta = eta;
      const double* zt = s_z + (slot * 2 + (i & 1)) * (kz + 1);
      pre_zl = (lane < kf) ? zt[lane] : 0.0;
      pre_Pj1 = lane_scan_wave(pre_zl * pre_zl);
      const double up = __shfl_up(pre_Pj1, 1, 64);
      pre_Pj = (lane == 0) ? 0.0 : up;
      pre_nrm2 = readlane_d(pre_Pj1, kf - 1);
    }
  };

#ifdef FMCMC_STAMP
  unsigned long long ow_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ow_prev = 0;
#define W2_OW_STAMP(i) do { if (wave == 0) { const unsigned long long t_ = stamp_clk(); ow_acc[i] += t_ - ow_prev; ow_prev = t_; } } while (0)
#else
#define W2_OW_STAMP(i) do { } while (0)
#endif
  publish(1u, true);                       // version 1: the initial state
  draw(2);
  wave_sync_lds();
  prepare(1);
#ifdef FMCMC_STAMP
  ow_prev = stamp_clk();
#endif
  bool ram_gate = false;                   // gate of the pending proposal (R/kernel_ram.R:129), decided when it was made
  bool lost = false;
  for (int v = 1; v <= nsteps; v++) {
    // ---- the lane partials of version v of this group are complete
    // (a wave without a chain waits and signals like the others: an arrival for version v + 1 must not come before every
    //  workgroup's arrival for version v)
    {
      const bool ok = w2_wait(X2, (unsigned)v * 8u, s_lost, (A.debug & 512) ? 400000u : 20000000u);
      if (!ok) { if (lane == 0) w2_lds_st(s_lost, 1u); lost = true; break; }
    }
    W2_OW_STAMP(0);
    // (measured and dropped, round 3: s_setprio 1 / 3 for the owner between here and the publish -- what gave C3's owners 6 % --
    //  is 18.5 us per step against 18.3: the evaluator wave that shares the SIMD is on the critical path as well)
    W2_EVENT(v == 300, 24 + 6 * g);                            // partials of version 300 seen (owner of group g)
    bool row_keep = false, fresh_prop = false;
    double row_th0 = 0.0, row_th1 = 0.0, row_f1 = 0.0;
    if (has && (status == FMCMC_CHAIN_OK)) {
      const double* pr = A.sh_part + (unsigned int)cl * (unsigned int)(NT + SH_PAD) + 8 * lane;
      // (eight consecutive partials per lane, 64-byte aligned: four 16-byte loads; an 8-byte sc1 access runs at 0.54-0.70 of
      //  the 16-byte rate)
      typedef double d2v_t __attribute__((ext_vector_type(2)));
      d2v_t q0, q1, q2, q3;
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                   "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                   "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                   "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pr) : "memory");
      const double v0 = q0.x, v1 = q0.y, v2 = q1.x, v3 = q1.y, v4 = q2.x, v5 = q2.y, v6 = q3.x, v7 = q3.y;
      const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));   // canonical levels 1, 2, 4 | 8 .. 256
      // closed form: -(n (log sigma + ln sqrt 2 pi)) - (tot / 2) / sigma^2 with the sigma-only part prepared while this wave
      // waited (same operations, same bits as finish_logpost)
      if (pre_sigma_ok) {
        f1 = -pre_nt1 - (0.5 * tot) / pre_ss;
        if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
      } else {
        f1 = finish_logpost<FMCMC_FAM_GAUSSIAN_LINREG>(A, L.th1, tot);
      }
      W2_OW_STAMP(1);
      row_th1 = (lane < k) ? L.th1[lane] : 0.0;
      bool do_update = false;                 // kernel_ram: (d_j, kappa_j) of this step are in CF
      if (v == 1) {
        f0 = f1;
        row_keep = true;
      } else {
        const int i = v;
        const double* zt = s_z + (slot * 2 + (i & 1)) * (kz + 1);
        if (KIND == FMCMC_KERNEL_RAM) {   // adaptation with f(theta1) of the (un-reflected == final) proposal, R/kernel_ram.R:129-152
          if (ram_gate) {
            double a_n = fmh_exp(f1 - f0);
            if (fmh_isnan(a_n)) a_n = 0.0;
            else if (a_n > 1.0) a_n = 1.0;
            const double cp = (pre_eta * (a_n - A.arate)) / pre_nrm2;
            if (cp != 0.0 && fmh_isfinite(cp)) {
              double dl, kl;
              const bool okl = ram_coef(cp, pre_Pj, pre_Pj1, pre_zl, dl, kl);
              if (__any(lane < kf && !okl)) {
                nerr += 1;
              } else {
                if (lane < kf) { CF[2 * lane] = dl; CF[2 * lane + 1] = kl; }
                do_update = true;
              }
            }
          }
          abs_iter += 1;
        }
        W2_OW_STAMP(2);
        // accept (R/mcmc.R:754-778); the row is stored behind the hand-over
        if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
        const double ratio = f1 - f0;
        if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
        if (status != FMCMC_CHAIN_OK) {
          if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
          if (lane < k) A.status_theta[cl * k + lane] = L.th1[lane];
        } else {
          const double lu = zt[kz];
          if (lu < ratio) {
            if (lane < k) L.th0[lane] = L.th1[lane];
            f0 = f1;
            nacc += 1;
            bitword |= (1u << ((i - 1) & 31));
          }
          row_keep = true;
        }
      }
      wave_sync_lds();
      row_th0 = (lane < k) ? L.th0[lane] : 0.0;
      row_f1 = f1;
      W2_OW_STAMP(3);
      // ---- factor update of step v and proposal of loop step v + 1 (a failed chain keeps its theta1: the table still holds it)
      if (status == FMCMC_CHAIN_OK && v < nsteps) {
        const int i = v + 1;
        const double* zt = s_z + (slot * 2 + (i & 1)) * (kz + 1);
        if (KIND == FMCMC_KERNEL_RAM) {   // R/kernel_ram.R:123-126
          const double s = (do_update && !(A.debug & 1024)) ? w2_ram_update_propose((lds_dptr_t)SG, (lds_dptr_t)CF, (lds_dptr_t)zt, LD, kf)   /* (1024: timing ablation, results invalid) */
                                     : w2_ram_propose((lds_dptr_t)SG, (lds_dptr_t)zt, LD, kf);
          if (lane < kf) {
            const int j = s_which[lane];
            L.th1[j] = L.th0[j] + s;
          }
          ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && (i % A.freq) == 0);
        } else {                          // kernel_normal(_reflective), joint scheme (R/kernel_normal.R:67-72, :159-164)
          if (lane < k) L.th1[lane] = L.th0[lane];
          wave_sync_lds();
          if (lane < kf) {
            const int j = s_which[lane];
            double t = L.th0[j] + (s_mu[j] + s_scale[j] * zt[lane]);
            if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_lb[j], s_ub[j]);
            L.th1[j] = t;
          }
        }
        wave_sync_lds();
        fresh_prop = true;
      } else if (KIND == FMCMC_KERNEL_RAM && do_update) {   // the last step of the call (or a chain that just failed): S only
        wave_sync_lds();
        w2_ram_update((lds_dptr_t)SG, (lds_dptr_t)CF, LD, kf);
        wave_sync_lds();
      }
    }
    W2_OW_STAMP(4);
    if (v < nsteps) publish((unsigned)(v + 1), fresh_prop);
    W2_OW_STAMP(5);
    W2_EVENT(v == 300, 25 + 6 * g);                            // version 301 published
    W2_EVENT(v == 299, 32 + g);                                // version 300 published
    // ---- everything below runs in the shadow of the hand-overs and of this group's matrix-core work
    if (row_keep) {   // row v of ans / draws / logpost (store_row reads L.th0 / L.th1: the saved values go through its registers)
      if (v > burnin) {
        thin_ctr += 1;
        if (thin_ctr == thin) {
          thin_ctr = 0;
          if (lane < k) {
            out_s[srow] = row_th0;
            if (out_d) out_d[srow] = row_th1;
          }
          if (out_l && lane == 0) out_l[srow] = row_f1;
          srow += 1;
        }
      }
    }
    if (has && v >= 2 && A.accept_bits && lane == 0 && (((v - 1) & 31) == 31 || v == nsteps)) {
      A.accept_bits[cl * (long long)((nsteps + 31) >> 5) + ((v - 1) >> 5)] = bitword;
      bitword = 0;
    }
    if (v < nsteps) {
      draw(v + 2);
      wave_sync_lds();
      prepare(v + 1);                     // what step v + 1 needs that does not depend on f(theta1)
      W2_OW_STAMP(6);
    }
  }
#ifdef FMCMC_STAMP
  if (wave == 0 && has && lane < 8 && k >= 32) A.status_theta[cl * k + lane] = (double)ow_acc[lane];
#endif
  // ---- write state back
  if (has) {
    if (lost) {
      status = FMCMC_CHAIN_SYNC_TIMEOUT;
      if (lane == 0) { A.status[cl] = status; A.status_step[cl] = 0; }
    }
    if (lane < k) A.theta0[cl * k + lane] = L.th0[lane];
    if (lane == 0) {
      A.f0[cl] = f0;
      A.accept_count[cl] = nacc;
      if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
      if (KIND == FMCMC_KERNEL_RAM) {
        A.abs_iter[cl] = abs_iter;
        if (A.nerrors) A.nerrors[cl] = nerr;
      }
    }
    if (KIND == FMCMC_KERNEL_RAM) {
      wave_sync_lds();
      for (int e = lane; e < kf * kf; e += 64) {
        const int a = e / kf, b = e % kf;
        A.Sigma[(cl * kf + a) * kf + b] = SG[2 * (a * LD + b)];
      }
    }
  }
}

}  // namespace
