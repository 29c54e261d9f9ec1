// msm_bred.cuh -- the bucket reduction  S = sum_k (k + 1) B_k  of the Pippenger MSM (the `running_sum` loop at the end of every window of
// upstream's multiexp_serial, halo2_proofs/src/arithmetic.rs @ v2023_04_20) as ONE launch of a radix-2 recursion.
//
// What it replaces (rounds 1-3): k_msm_reduce_local (a quad per 4 / 8 buckets: local running sums, then k0 * run by double-and-add -- a chain of
// ~33 dependent group operations, ~6.5 group operations per bucket) + two k_msm_tree_sum launches.
//
// The recursion: write k = sum_j 2^j k_j.  Then  sum_k k B_k = sum_j 2^j A_j  with  A_j = sum of the buckets whose index has bit j set, and all A_j
// come out of ONE binary tree over the buckets: a node covering 2^t buckets carries the vector (X, A_0 .. A_{t-1}) -- its total and its bit sums,
// indices taken relative to the node -- and two siblings combine as
//       X = X_lo + X_hi,    A_j = A_j(lo) + A_j(hi)  (j < t),    A_t = X_hi            (no addition: the high half IS the set with bit t),
// t + 1 independent additions per node.  2 N additions for N buckets (against ~6.5 N), no doubling until the very end
// (S = X + sum_j 2^j A_j: j doublings of A_j, side by side), every level's additions independent of each other.
//
// Mapping: a 256-thread block = 64 quads takes 256 consecutive buckets into LDS (the accumulation kernel leaves the LDS unused) and works in place;
// every addition is QUAD-cooperative with its operands IN MEMORY (x29q_add_mem: a lane loads only the coordinate its multiplication needs and stores
// only the coordinate it produced -- no 36-word operands in registers, no select chains), which keeps the kernel below 128 VGPRs: a wave of it fits
// beside three resident waves of k_msm_accum0 on a SIMD.  Blocks leave their node vector (<= 9 points) in HBM; the LAST block of a cluster of 16
// (an atomic counter per cluster, __threadfence on both sides) combines the cluster's 16 vectors, the last cluster of a group the <= 8 cluster
// vectors, and that same block weights, sums and emits the result: one launch per MSM batch instead of three.
#pragma once
#include "ec29.cuh"

// the measurement hooks (wall-clock stamps of the phases: DEHALO_MSM_BRED_STAMPS, DEHALO_MSM_MERGE_STAMPS, DEHALO_MSM_MERGE_Q3) exist in -DDEHALO_EXPERIMENTS builds only
#ifndef DEHALO_PHASE_STAMPS
#ifdef DEHALO_EXPERIMENTS
#define DEHALO_PHASE_STAMPS 1
#else
#define DEHALO_PHASE_STAMPS 0
#endif
#endif
#define BRED_THREADS 256
#define BRED_QUADS (BRED_THREADS / 4)
#define BRED_BLOCK_BUCKETS 256      // buckets of a block at most (the LDS array): 4 per quad
#define BRED_BLOCK_BUCKETS_MIN 128  // ... and at least (sizes the node buffers); run_msm_t picks (msm_bred_block)
#define BRED_FANIN 16               // node vectors one block combines in the second / third stage
#define BRED_VMAX 17                // points per node vector in HBM: A_0 .. A_15, X
#define BRED_CNT_PER_GROUP 32       // u32 counters per bucket group: clusters [0, 16), group [16]

// ---- quad-cooperative group operations on records in memory (LDS or HBM): record = 36 words, coordinate c at words [9 c, 9 c + 9) ----
FP_DEV f29 q_ld(const u32* rec, u32 coord) {
    f29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = rec[9 * coord + i];
    return r;
}
FP_DEV void q_st(u32* rec, u32 coord, const f29& v) {
#pragma unroll
    for (int i = 0; i < 9; i++) rec[9 * coord + i] = v.v[i];
}
FP_DEV bool q_is_literal_identity(const u32* rec) {      // ZZ = 0 limb by limb: an empty bucket, x29_identity()
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= rec[18 + i];
    return o == 0;
}
template <int K>
FP_DEV u32 q_bcast_word(u32 a) {
    u32 t = (u32)__builtin_amdgcn_update_dpp(0, (int)a, K * 0x55, 0xf, 0xf, false);
    asm volatile("" : "+v"(t));
    return t;
}
FP_DEV void x29q_copy(u32* out, const u32* src, u32 role) {
    if (out != src) q_st(out, role, q_ld(src, role));
}

// out = 2 a  (all four lanes of the quad active, same pointers; out may be a)
template <class F>
FP_DEV void x29q_double_mem(u32* out, const u32* a) {
    const u32 role = threadIdx.x & 3;
    if (q_is_literal_identity(a)) { x29q_copy(out, a, role); return; }
    const f29 ax = q_ld(a, 0), ay = q_ld(a, 1);
    const f29 u = f29_dbl(ay);
    // level 1: v = U^2 | xx = X^2
    f29 op = (role & 1) ? ax : u;
    f29 m = f29_mul<F>(op, op);
    const f29 v = f29_quad_bcast<0>(m), xx = f29_quad_bcast<1>(m);
    const f29 mm_in = f29_norm(f29_add(f29_dbl(xx), xx));
    // level 2: w = U V | s = X V | mm = M^2 | zz3 = V ZZ
    m = f29_mul<F>(f29_sel4(u, ax, mm_in, v, role), f29_sel4(v, v, mm_in, q_ld(a, 2), role));
    const f29 w = f29_quad_bcast<0>(m), s = f29_quad_bcast<1>(m), mm = f29_quad_bcast<2>(m);
    const f29 m2 = m;                                   // lane 3 keeps zz3
    const f29 rx = f29_norm(f29_sub(mm, f29_dbl(s), F::KB));
    const f29 t = f29_sub(s, rx, F::KA);
    // level 3: M T | W Y | zzz3 = W ZZZ | (lane 3: spare)
    m = f29_mul<F>(f29_sel4(mm_in, w, w, w, role), f29_sel4(t, ay, q_ld(a, 3), ay, role));
    const f29 mt = f29_quad_bcast<0>(m), wy = f29_quad_bcast<1>(m);
    const f29 ry = f29_norm(f29_sub(mt, wy, F::KM));
    // lane 0 stores x, lane 1 y, lane 3 zz (its level-2 product), lane 2 zzz (its level-3 product)
    q_st(out, role < 2 ? role : 5 - role, f29_sel4(rx, ry, m, m2, role));
}

// out = a + b  (all four lanes of the quad active, same pointers; out may be a or b).  Same arithmetic, level by level, as x29_add_quad (ec29.cuh).
template <class F>
FP_DEV void x29q_add_mem(u32* out, const u32* a, const u32* b) {
    const u32 role = threadIdx.x & 3;
    if (q_is_literal_identity(a)) { x29q_copy(out, b, role); return; }
    if (q_is_literal_identity(b)) { x29q_copy(out, a, role); return; }
    // level 1: u1 = X1 ZZ2 | u2 = X2 ZZ1 | s1 = Y1 ZZZ2 | s2 = Y2 ZZZ1
    const u32* pa = (role & 1) ? b : a;
    const u32* pb = (role & 1) ? a : b;
    f29 m = f29_mul<F>(q_ld(pa, role >> 1), q_ld(pb, 2 + (role >> 1)));
    const f29 u1 = f29_quad_bcast<0>(m), u2 = f29_quad_bcast<1>(m), s1 = f29_quad_bcast<2>(m), s2 = f29_quad_bcast<3>(m);
    const f29 p = f29_norm(f29_sub(u2, u1, F::KM));
    const f29 rr = f29_norm(f29_sub(s2, s1, F::KM));
    // level 2: pp = P^2 | r2 = R^2 | zz12 = ZZ1 ZZ2 | zzz12 = ZZZ1 ZZZ2   (lanes 2 / 3 load their coordinate of both operands)
    {
        const f29 la = q_ld(a, role), lb = q_ld(b, role);
        f29 oa, ob;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const u32 sq = (role & 1) ? rr.v[i] : p.v[i];
            oa.v[i] = (role & 2) ? la.v[i] : sq;
            ob.v[i] = (role & 2) ? lb.v[i] : sq;
        }
        m = f29_mul<F>(oa, ob);
    }
    const f29 pp = f29_quad_bcast<0>(m), r2 = f29_quad_bcast<1>(m);
    const f29 keep = m;                                 // lanes 2 / 3 keep zz12 / zzz12
    // level 3: ppp = P PP | qq = U1 PP | zz3 = zz12 PP | (lane 3: spare)
    m = f29_mul<F>(f29_sel4(p, u1, keep, p, role), pp);
    const f29 ppp = f29_quad_bcast<0>(m), qq = f29_quad_bcast<1>(m);
    const f29 m3 = m;                                   // lane 2 keeps zz3
    const u32 zz3_0 = q_bcast_word<2>(m.v[0]);
    const f29 rx = f29_norm(f29_sub(r2, f29_add(ppp, f29_dbl(qq)), F::KB));
    const f29 t = f29_sub(qq, rx, F::KA);
    // level 4: R T | S1 PPP | (lane 2: spare) | zzz3 = zzz12 PPP
    m = f29_mul<F>(f29_sel4(rr, s1, rr, keep, role), f29_sel4(t, ppp, t, ppp, role));
    const f29 rt = f29_quad_bcast<0>(m), sp = f29_quad_bcast<1>(m);
    const f29 ry = f29_norm(f29_sub(rt, sp, F::KM));
    if (__builtin_expect((zz3_0 == 0) | (zz3_0 == F::P[0]), 0)) {
        if (f29_is_zero_lt2p<F>(f29_quad_bcast<2>(m3))) {
            // exceptional (ZZ3 = 0 mod p): an identity operand that is not all-zero limbs, P == Q, or P == -Q -- decided exactly, every lane alike
            if (f29_is_zero_slow<F>(q_ld(a, 2))) x29q_copy(out, b, role);
            else if (f29_is_zero_slow<F>(q_ld(b, 2))) x29q_copy(out, a, role);
            else if (f29_is_zero_slow<F>(rr)) x29q_double_mem<F>(out, a);
            else q_st(out, role, f29_zero());
            return;
        }
    }
    q_st(out, role, f29_sel4(rx, ry, m3, m, role));
}

// ---- the tree --------------------------------------------------------------------------------------------------------------
// K node vectors in LDS, vector i with C base components at records aBase + i * iStride + c (c < C) and its total at xBase + i * iStride.
// In place: after the call vector 0 holds the combined node -- base components where they were, the log2 K new ones A_{C + s} in the TOTAL slot
// of vector 2^s, the total in the total slot of vector 0.  (With C = 0 and K raw buckets this is the whole recursion from the leaves up; the same
// call with C = 0 over any K records also just sums them into record 0.)  ONE inlined copy of the addition per kernel: the instruction cache holds
// 64 KB and an addition is ~16 KB of code -- the first version of this kernel inlined it ten times and ran slower than what it replaced.
template <class F>
FP_DEV void bred_tree(u32* lds, u32 aBase, u32 xBase, u32 iStride, u32 C, u32 K) {
    const u32 quad = threadIdx.x >> 2;
    for (u32 t = 0; (1u << t) < K; t++) {
        const u32 per = C + 1 + t, jobs = (K >> (t + 1)) * per;
        for (u32 job = quad; job < jobs; job += BRED_QUADS) {
            const u32 node = job / per, comp = job - node * per;
            const u32 lo = node << (t + 1), hi = lo + (1u << t);
            u32 d, s;
            if (comp < C) { d = aBase + lo * iStride + comp; s = aBase + hi * iStride + comp; }
            else if (comp == C) { d = xBase + lo * iStride; s = xBase + hi * iStride; }
            else { const u32 o = 1u << (comp - C - 1); d = xBase + (lo + o) * iStride; s = xBase + (hi + o) * iStride; }
            x29q_add_mem<F>(lds + 36 * d, lds + 36 * d, lds + 36 * s);
        }
        __syncthreads();
    }
}

// the combined vector of bred_tree, in canonical order [A_0 .. A_{C + log2 K - 1}, X], copied to `dst` (LDS records)
FP_DEV void bred_gather(const u32* lds, u32 aBase, u32 xBase, u32 iStride, u32 C, u32 K, u32* dst) {
    u32 logK = 0;
    while ((1u << logK) < K) logK++;
    const u32 V = C + logK + 1;
    for (u32 e = threadIdx.x; e < V * 36; e += BRED_THREADS) {
        const u32 v = e / 36, w = e - v * 36;
        const u32 src = v < C ? aBase + v : (v < C + logK ? xBase + (1u << (v - C)) * iStride : xBase);
        dst[e] = lds[36 * src + w];
    }
}

// One thread block per bb (= 128 or 256) buckets of a group; grid (max(1, nb / bb), total_groups).  (128: the block's tree is 7 rounds of additions instead of 11
// -- a level with fewer nodes than quads costs a round all the same -- for one more level further up.)  nodes1 / nodes2: BRED_VMAX records per block / per cluster.
// counters: BRED_CNT_PER_GROUP words per group, zero on entry and left zero.  The result of group g goes to fin_out / fin_affine [g] (precomputed
// tables: one group per MSM) or, when both are null, to gsums[g] for k_msm_final.
// Phases of a block: 0 its 256 buckets | 1 (last block of a cluster of 16) the cluster's vectors | 2 (last cluster of the group) the clusters' vectors |
// 3 (the block that holds the group's vector) weighting by 2^j and the final sum.  Every phase is one call of the tree -- one loop body.
// measurement only (DEHALO_MSM_BRED_STAMPS=1): wall-clock stamps (100 MHz) of the block that finishes group 0 -- [0] its start, [1 + phase] the end of each phase's
// tree, [5] the doublings of phase 3 done, [6] the result written; [7] the earliest start of any block; the hand-offs of phases 1 / 2: [8] / [10] this block knows it
// is the last arrival (its vector published, fence, counter), [9] / [11] the siblings' vectors are in LDS
#if DEHALO_PHASE_STAMPS
__device__ unsigned long long g_bred_stamps[12];
__device__ int g_bred_stamps_on;
#define BRED_STAMPS_ON (g_bred_stamps_on != 0)
#define BRED_STAMP(i) do { if (stamps_on && tid == 0 && g == 0) my_stamps[i] = wall_clock64(); } while (0)
#else
#define BRED_STAMPS_ON false
#define BRED_STAMP(i) do { } while (0)
#endif

template <class CV>
__global__ __launch_bounds__(BRED_THREADS) void k_msm_bred(u32 nb, u32 bb, const xyzz29_rec* buckets, xyzz29_rec* nodes1, xyzz29_rec* nodes2, u32* counters, xyzz29_rec* gsums,
                                                          jacobian_t* fin_out, affine_t* fin_affine) {
    msm_tail_prio();
    typedef f29_lat<typename f29_of<typename CV::Base>::type> F;      // lone waves of dependent operations: the latency schedule (tools/ubench_qmem.hip: 2.9 against 3.5 us per addition, 78 VGPRs either way)
    __shared__ __align__(16) u32 lds[(BRED_BLOCK_BUCKETS + 32) * 36];
    __shared__ u32 s_last;
    u32* const vec = lds + BRED_BLOCK_BUCKETS * 36;      // 32 records: the node vector being handed on
    const u32 tid = threadIdx.x, quad = tid >> 2;
    const u32 g = blockIdx.y;
    const u32 here = nb < bb ? nb : bb;      // buckets of a block (a power of two >= 8)
    u32 m = 0;                                                               // log2 nb
    while ((1u << m) < nb) m++;
#if DEHALO_PHASE_STAMPS
    const bool stamps_on = BRED_STAMPS_ON;
    unsigned long long my_stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    BRED_STAMP(0);
    if (stamps_on && tid == 0 && g == 0) atomicMin(&g_bred_stamps[7], my_stamps[0]);
#endif

    u32 children = gridDim.x;      // vectors still to combine for this group after the current phase
    u32 idx = blockIdx.x;          // this block's position among them
    u32 V = 0;                     // records of the vector in `vec`
    for (u32 phase = 0; phase < 4; phase++) {
        u32 C, K, xBase, iStride;
        if (phase == 0) {          // the block's buckets -> LDS (plain records: a leaf is its own total)
            const uint4* src = reinterpret_cast<const uint4*>(buckets + (u64)g * nb + (u64)idx * bb);
            uint4* dst = reinterpret_cast<uint4*>(lds);
            for (u32 e = tid; e < here * 9; e += BRED_THREADS) dst[e] = src[e];
            C = 0; K = here; xBase = 0; iStride = 1;
        } else if (phase < 3) {    // publish this block's vector; the last arrival of its cluster goes on with the cluster's vectors
            if (children == 1) continue;
            xyzz29_rec* level_nodes = phase == 1 ? nodes1 : nodes2;
            const u32 fan = children < BRED_FANIN ? children : BRED_FANIN;
            const u32 cl = idx / fan;
            u32* mine = reinterpret_cast<u32*>(level_nodes + ((u64)g * children + idx) * BRED_VMAX);
            for (u32 e = tid; e < V * 36; e += BRED_THREADS) mine[e] = vec[e];
            // ONE device-scope fence per block on either side (each is a write-back / invalidation of the XCD's whole L2: with one per wave -- 2,560 of them for
            // ten columns -- the kernel spent more time in them than in its additions).  The block barrier's workgroup-scope release has every wave's stores in
            // the L2 before thread 0 writes it back; the invalidation by thread 0 serves the whole block (one CU, one L1).
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                u32* cnt = counters + (u64)g * BRED_CNT_PER_GROUP + (phase == 1 ? cl : 16);
                const u32 old = atomicAdd(cnt, 1u);
                s_last = old == fan - 1;
                if (s_last) {
                    *cnt = 0;                              // (everyone else has counted already: left zero for the next launch)
                    __threadfence();
                }
            }
            __syncthreads();
            if (!s_last) return;
            BRED_STAMP(6 + 2 * phase);
            const u32* kids = reinterpret_cast<const u32*>(level_nodes + ((u64)g * children + (u64)cl * fan) * BRED_VMAX);
            for (u32 e = tid; e < fan * V * 36; e += BRED_THREADS) {
                const u32 i = e / (V * 36), w = e - i * (V * 36);
                lds[e] = kids[(u64)i * BRED_VMAX * 36 + w];
            }
            C = V - 1; K = fan; xBase = V - 1; iStride = V;
            children /= fan; idx = cl;
#if DEHALO_PHASE_STAMPS
            __syncthreads();
            BRED_STAMP(7 + 2 * phase);
#endif
        } else {                   // vec = [A_0 .. A_{m-1}, X]  ->  S = X + sum_j 2^j A_j: quad j doubles A_j j times, then the 16 records are summed
            const u32 slots = m + 1 <= 16 ? 16u : 32u;      // (2^16 buckets -- a 17-bit window --: 17 records, one more level of the final tree)
            for (u32 e = tid; e < slots * 36; e += BRED_THREADS) lds[e] = e < (m + 1) * 36 ? vec[e] : 0u;
            __syncthreads();
            if (quad < m)
                for (u32 i = 0; i < quad; i++) x29q_double_mem<F>(lds + 36 * quad, lds + 36 * quad);
            C = 0; K = slots; xBase = 0; iStride = 1;
        }
        __syncthreads();
        if (phase == 3) BRED_STAMP(5);
        bred_tree<F>(lds, 0, xBase, iStride, C, K);
        BRED_STAMP(1 + phase);
        if (phase < 3) {
            bred_gather(lds, 0, xBase, iStride, C, K, vec);
            u32 lk = 0;
            while ((1u << lk) < K) lk++;
            V = C + lk + 1;
            __syncthreads();
        }
    }
    if (tid == 0) {
        const xyzz29 s = x29_load(reinterpret_cast<const xyzz29_rec*>(lds));
        if (fin_out || fin_affine) msm_emit<F>(s, fin_out ? fin_out + g : nullptr, fin_affine ? fin_affine + g : nullptr);
        else x29_store(&gsums[g], s);
#if DEHALO_PHASE_STAMPS
        if (stamps_on && g == 0) {
            my_stamps[6] = wall_clock64();
            for (int i = 0; i < 12; i++) if (i != 7) g_bred_stamps[i] = my_stamps[i];
        }
#endif
    }
}

// ==========================================================================================================================================
// Merging a bucket's partial sums with operands in LDS (round 4): the same classes and sections as k_msm_merge_all (msm.cuh), every addition
// quad-cooperative on records in LDS (x29q_add_mem) -- below 128 VGPRs, so that a wave of it fits beside three resident waves of k_msm_accum0.
// A quad keeps its running sum in LDS record 2 q and stages the next partial sum (read from HBM one iteration ahead, nine words per lane) in
// record 2 q + 1.  All lanes of a quad belong to one wave: LDS operations of a wave complete in order, so a record written coordinate by
// coordinate by the four lanes is complete for every one of them at the next instruction; __builtin_amdgcn_wave_barrier() keeps the compiler from
// moving LDS accesses across those points.
FP_DEV void q_copy_in(u32* rec, const xyzz29_rec* src, u32 role) {      // HBM record -> LDS record, nine words per lane
    const u32* s = reinterpret_cast<const u32*>(src) + 9 * role;
#pragma unroll
    for (int i = 0; i < 9; i++) rec[9 * role + i] = s[i];
}
FP_DEV void q_copy_out(xyzz29_rec* dst, const u32* rec, u32 role) {
    u32* d = reinterpret_cast<u32*>(dst) + 9 * role;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = rec[9 * role + i];
}

// acc (LDS record) = sum of partial[p], p = first, first + step, ... < end  (identity if none); the next record is in flight during an addition
#if DEHALO_PHASE_STAMPS
__device__ unsigned long long g_merge2_iter[64];      // measurement only: the iterations of one quad's walk (first block of class 3, quad 0)
#endif
template <class F>
FP_DEV void q_strided_sum(u32* acc, u32* inc, const xyzz29_rec* partial, u32 first, u32 step, u32 end, u32 role, unsigned long long* iter_stamps = nullptr) {
    if (first >= end) {
#pragma unroll
        for (int i = 0; i < 9; i++) acc[9 * role + i] = 0;
        __builtin_amdgcn_wave_barrier();
        return;
    }
    q_copy_in(acc, &partial[first], role);
    u32 nxt[9];
    u32 p = first + step;
    if (p < end) {
        const u32* s = reinterpret_cast<const u32*>(&partial[p]) + 9 * role;
#pragma unroll
        for (int i = 0; i < 9; i++) nxt[i] = s[i];
    }
    u32 it = 0;
    if (iter_stamps) iter_stamps[it++] = wall_clock64();
    while (p < end) {
        if (iter_stamps && it < 60) iter_stamps[it++] = wall_clock64();
#pragma unroll
        for (int i = 0; i < 9; i++) inc[9 * role + i] = nxt[i];
        p += step;
        if (p < end) {
            const u32* s = reinterpret_cast<const u32*>(&partial[p]) + 9 * role;
#pragma unroll
            for (int i = 0; i < 9; i++) nxt[i] = s[i];
        }
        __builtin_amdgcn_wave_barrier();
        x29q_add_mem<F>(acc, acc, inc);
        __builtin_amdgcn_wave_barrier();
    }
    if (iter_stamps && it < 62) { iter_stamps[it++] = wall_clock64(); iter_stamps[it] = 0; }
    __builtin_amdgcn_wave_barrier();
}

#define MERGE2_BLOCKS_LIGHT 1024     // per light class
#define MERGE2_BLOCKS_Q8 512
#define MERGE2_BLOCKS_BLOCK 512
#define MERGE2_BLOCKS_PARTS 1024
#define MERGE2_BLOCKS_COPY 256        // buckets with one partial sum (copied) or none (identity): one lane per bucket, the last section of the grid
#define MERGE2_GRID_SUMS (MERGE2_BLOCKS_PARTS + MERGE2_BLOCKS_BLOCK + MERGE2_BLOCKS_Q8 + 3 * MERGE2_BLOCKS_LIGHT)
#define MERGE2_GRID (MERGE2_GRID_SUMS + MERGE2_BLOCKS_COPY)

// classification for k_msm_merge2: one lane per bucket; S = 0 -> identity, S = 1 -> copy, otherwise the bucket is queued in the list of its class
// (one atomic per wave and class).  The classes are cut so that no chain is longer than ~14 additions whatever S is:
//   0 | 1 | 2   S = 2 | 3-4 | 5-8: one quad walks the bucket's records (three lists, three grid sections side by side: a wave of 16 quads runs as long as its
//               longest bucket, and sections that ran one after the other inside a block added their latencies up -- 80 us for the trivial merge of a K = 11 proof);
//   3           S = 9 .. 64: 8 quads (strided quad sums, then a tree inside the wave);
//   4           S = 65 .. 512: a block of 64 quads (LDS tree);
//   5 / 6       S > 512: the bucket is cut into parts of 512 records, list 5 holds one entry (slot, part) per part, list 6 one entry (bucket, first part, parts,
//               arrival counter) per such bucket; a block sums one part into the parts buffer and the LAST block of a bucket to arrive sums its parts.  (One block per
//               bucket, as before round 4, walked S / 64 records per quad: 0.3 ms for the 10^4-record buckets of a permuted lookup column.)
// (the lists are built by k_scan_offsets, msm.cuh, from the record ranges alone -- before the accumulation runs; until round 4 a launch of its own between the
// accumulation and this kernel, k_msm_merge_classify2, which also copied the S <= 1 buckets: the last grid section below does that now)

// grid = [parts of heavy buckets | block class | 8-quad class | light 5-8 | light 3-4 | light 2] sections of 256-thread blocks; a block whose section's list is
// shorter than its position leaves at once.  One loop body serves all classes (Q = 64 / 64 / 8 / 1 quads per unit: strided quad sums, then a tree over the
// Q quads), so that the addition is inlined twice, not five times (instruction cache).
// (Round 4, measured and removed: summing a populous class -- every bucket of a dense MSM in one class, 16384 x ~15 or 32768 x ~13 partial sums -- by plain lanes,
// T lanes per bucket with register operands and shuffle folds, instead of quads.  A lane's own addition is 14 multiplications one after the other, ~3,800
// instructions against ~1,300 for the quad-cooperative one, so a wave-step takes 8-16 us against 3.6 and the four-fold lane economy is spent: the kernel took
// ~95 us where the quads take 100 at 2^17 and the whole MSM got 20-70 us longer; a k = 17 proof +0.1-0.2 ms.  profiles/r04_merge_lanes_ab.txt)
// measurement only (DEHALO_MSM_MERGE_STAMPS=1): per block [start, its section's count read, end] wall-clock stamps (100 MHz) and [class, Q, units it summed]
#if DEHALO_PHASE_STAMPS
__device__ unsigned long long g_merge2_stamps[MERGE2_GRID * 3];
__device__ u32 g_merge2_info[MERGE2_GRID * 3];
__device__ int g_merge2_stamps_on;
__device__ int g_merge2_q3;
#define MERGE2_STAMP(i) do { if (stamps_on && tid == 0) g_merge2_stamps[3 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define MERGE2_STAMP(i) do { } while (0)
#endif

template <class CV>
__global__ __launch_bounds__(256) void k_msm_merge2(const u32* rbeg, const u32* rend, const xyzz29_rec* partial, xyzz29_rec* buckets, const u32* counters, u32* lists, u32 cap,
                                                   xyzz29_rec* parts_buf, u32 total_buckets) {
    msm_tail_prio();
    typedef f29_lat<typename f29_of<typename CV::Base>::type> F;
    __shared__ __align__(16) u32 lds[128 * 36];
    __shared__ u32 s_last;
    const u32 tid = threadIdx.x, quad = tid >> 2, role = tid & 3;
    u32* acc = lds + 36 * (2 * quad);
    u32* inc = acc + 36;
    u32 blk = blockIdx.x, cls, nblk, Q;
    if (blk >= MERGE2_GRID_SUMS) {                           // S <= 1: nothing to add
        for (u32 b = (blk - MERGE2_GRID_SUMS) * 256 + tid; b < total_buckets; b += MERGE2_BLOCKS_COPY * 256) {
            const u32 beg = rbeg[b], S = rend[b] - beg;
            if (S > 1) continue;
            xyzz29_rec rec;
            if (S == 1) rec = partial[beg];
            else {
#pragma unroll
                for (int i = 0; i < 36; i++) rec.w[i] = 0;
            }
            buckets[b] = rec;
        }
        return;
    }
    if (blk < MERGE2_BLOCKS_PARTS) { cls = 5; nblk = MERGE2_BLOCKS_PARTS; Q = 64; }
    else if ((blk -= MERGE2_BLOCKS_PARTS) < MERGE2_BLOCKS_BLOCK) { cls = 4; nblk = MERGE2_BLOCKS_BLOCK; Q = 64; }
    else if ((blk -= MERGE2_BLOCKS_BLOCK) < MERGE2_BLOCKS_Q8) { cls = 3; nblk = MERGE2_BLOCKS_Q8; Q = 8; }
    else { blk -= MERGE2_BLOCKS_Q8; cls = 2 - blk / MERGE2_BLOCKS_LIGHT; blk %= MERGE2_BLOCKS_LIGHT; nblk = MERGE2_BLOCKS_LIGHT; Q = 1; }
#if DEHALO_PHASE_STAMPS
    const bool stamps_on = g_merge2_stamps_on != 0;
#endif
    MERGE2_STAMP(0);
    const u32 count = counters[cls];
    MERGE2_STAMP(1);
    u32 units = 0;
    // Wide groups buy latency with idle lanes (a tree level keeps half of the group's quads busy): right for the few hundred skewed buckets of a witness column,
    // wrong when EVERY bucket of a large dense MSM lands in the class (2^20 uniform scalars: 32768 buckets of ~13 records -- eight sweeps of 8-quad groups
    // against one sweep of single quads, step 1.37 -> 1.50 ms).  So a populous class falls back to narrower groups.
    // Class 3 takes the widest group that still gives every bucket its group in ONE sweep of the section (512 blocks x 64 quads): 8 quads up to 4096
    // buckets, 4 up to 8192, 2 up to 16384 (three 2^14 columns: 12288 buckets of ~20 records, 200 -> 150 us; a lone 2^17 column 106 -> 99), single quads beyond (2^20).
    // What bounds a populous class is issue slots, not latency: a lone wave of quad additions takes 3.6-3.8 us per step here and fills its SIMD; two per SIMD take 7.
    if (cls == 3) {
        static_assert(MERGE2_BLOCKS_Q8 == 512, "");
#if DEHALO_PHASE_STAMPS
        if (g_merge2_q3) Q = count > MERGE2_BLOCKS_Q8 * 8 * 2 ? 1 : 8;            // (DEHALO_MSM_MERGE_Q3=1: the two-way choice of the first version, for the A/B)
        else
#endif
        while (Q > 1 && count * Q > MERGE2_BLOCKS_Q8 * 64) Q >>= 1;
    }
    if (cls == 4 && count > MERGE2_BLOCKS_BLOCK * 4) Q = 8;
    const bool wide = Q == 64;                               // quads of one unit span several waves: block barriers (these loops are uniform over the block)
    const u32 per_block = 64 / Q, grp = quad / Q, q = quad % Q;
    const u32* list = lists + cls * (size_t)cap;
    for (u32 i = blk * per_block + grp; i < count; i += nblk * per_block) {
        const xyzz29_rec* src = partial;
        xyzz29_rec* dst;
        u32 beg, end;
        u32* hb = nullptr;
        if (cls == 5) {                                      // one part of a heavy bucket -> the parts buffer
            hb = lists + (size_t)6 * cap + 4 * (size_t)list[2 * i];
            const u32 part = list[2 * i + 1], b = hb[0];
            beg = rbeg[b] + part * MERGE2_CHUNK;
            end = min(rend[b], beg + MERGE2_CHUNK);
            dst = &parts_buf[hb[1] + part];
        } else {
            const u32 b = list[i];
            beg = rbeg[b]; end = rend[b];
            dst = &buckets[b];
        }
        for (u32 round = 0; round < 2; round++) {            // (round 1: only the last block of a heavy bucket, over the bucket's parts)
#if DEHALO_PHASE_STAMPS
            q_strided_sum<F>(acc, inc, src, beg + q, Q, end, role, stamps_on && tid == 0 && blockIdx.x == MERGE2_BLOCKS_PARTS + MERGE2_BLOCKS_BLOCK && units == 0 ? g_merge2_iter : nullptr);
#else
            q_strided_sum<F>(acc, inc, src, beg + q, Q, end, role);
#endif
            if (wide) __syncthreads();
            for (u32 d = Q >> 1; d >= 1; d >>= 1) {
                if (q < d) x29q_add_mem<F>(acc, acc, acc + 72 * d);
                if (wide) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            }
            if (q == 0) q_copy_out(dst, acc, role);
            if (wide) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            if (cls != 5 || round == 1) break;
            // the part is in HBM: count this block in; the last one to arrive sums the bucket's parts (one device-scope fence per block on either side, msm_bred above)
            if (tid == 0) {
                __threadfence();
                s_last = atomicAdd(&hb[3], 1u) == hb[2] - 1;
                if (s_last) __threadfence();
            }
            __syncthreads();
            if (!s_last) break;
            src = parts_buf; beg = hb[1]; end = hb[1] + hb[2]; dst = &buckets[hb[0]];
        }
        units++;
    }
#if DEHALO_PHASE_STAMPS
    if (stamps_on && tid == 0) {
        g_merge2_stamps[3 * blockIdx.x + 2] = wall_clock64();
        g_merge2_info[3 * blockIdx.x] = cls; g_merge2_info[3 * blockIdx.x + 1] = Q; g_merge2_info[3 * blockIdx.x + 2] = units;
    }
#endif
    (void)units;
}

#if DEHALO_PHASE_STAMPS
// measurement only: what the stamps of the last k_msm_merge2 launch say (stderr)
static inline int merge2_report_stamps(dehalo_ctx* ctx, u32 tb, hipStream_t s) {
    std::vector<unsigned long long> st(MERGE2_GRID * 3); std::vector<u32> info(MERGE2_GRID * 3);
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_merge2_stamps), st.size() * 8));
    HIP_TRY(ctx, hipMemcpyFromSymbol(info.data(), HIP_SYMBOL(g_merge2_info), info.size() * 4));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (u32 b = 0; b < MERGE2_GRID_SUMS; b++) { t0 = std::min(t0, st[3 * b]); t1 = std::max(t1, st[3 * b + 2]); }
    fprintf(stderr, "k_msm_merge2 %u buckets: %.1f us from the first block's start to the last block's end;", tb, (double)(t1 - t0) / 100.0);
    for (u32 c = 0; c < 6; c++) {
        u32 nb_ = 0, q_ = 0, umax = 0; unsigned long long usum = 0; double smax = 0, cmax = 0, emax = 0, longest = 0;
        for (u32 b = 0; b < MERGE2_GRID_SUMS; b++) {
            if (info[3 * b] != c || info[3 * b + 2] == 0) continue;
            nb_++; q_ = info[3 * b + 1]; umax = std::max(umax, info[3 * b + 2]); usum += info[3 * b + 2];
            smax = std::max(smax, (double)(st[3 * b] - t0) / 100.0); cmax = std::max(cmax, (double)(st[3 * b + 1] - t0) / 100.0);
            emax = std::max(emax, (double)(st[3 * b + 2] - t0) / 100.0); longest = std::max(longest, (double)(st[3 * b + 2] - st[3 * b + 1]) / 100.0);
        }
        if (nb_) fprintf(stderr, " class %u: Q %u, %u blocks with work (%llu units, <= %u per block), latest start %.1f, latest count read %.1f, latest end %.1f, longest block %.1f us;",
                         c, q_, nb_, usum, umax, smax, cmax, emax, longest);
    }
    unsigned long long it[64];
    HIP_TRY(ctx, hipMemcpyFromSymbol(it, HIP_SYMBOL(g_merge2_iter), sizeof(it)));
    fprintf(stderr, " | one quad's walk of class 3, us per iteration:");
    for (int i = 1; i < 62 && it[i]; i++) fprintf(stderr, " %.1f", (double)(it[i] - it[i - 1]) / 100.0);
    fprintf(stderr, "\n");
    return 0;
}
#endif
