// iba_factor2_kernel (round 6): the residual blocks of a batch's work lists -> robust normal equations, as EQUAL SHARES OF A CANDIDATE'S WHOLE
// LIST instead of one wave per (keyframe, candidate).
//
// What rounds 2-5 ran (iba_factor_kernel, iba_kernels.hpp) gives every (keyframe, candidate) a one-wave block: ~264 list entries, i.e. ~130
// plane factors and ~130 3d-3d factors, worked off 64 at a time: two full passes and a third with two lanes alive, per kind. A third of the
// f64 instructions a block issues belong to lanes that are switched off, its 41 sums are cleared and reduced through LDS for ~4 blocks per lane,
// and the block starts with a chain of four dependent round trips (candidate, keyframe header -> relative poses -> list -> gathers) that
// two waves per SIMD cannot hide: 122 us, VALU-active 0.59 (profiles/r05j).
//
// Here a candidate's lists (its keyframes' rows of the work list, concatenated in keyframe order) are cut into W equal ranges of ENTRIES, W x B ~ the
// wave slots of the device (2 per SIMD), one wave per range:
//   * a wave walks ~T / W entries across keyframe boundaries; the entries are split, as before, into dense queues per kind of block (plane factor /
//     point-to-plane / point-to-point) and a body runs whenever its queue holds 64: every pass is full but the last of a wave, whatever the keyframes'
//     list lengths are;
//   * a queue item carries its keyframe; what a body needs of a keyframe (camera, pose, offsets of its tables, the relative poses of its covisible
//     slots with the translation pre-multiplied by s) sits in a small LDS ring, written when the walk reaches the keyframe from registers that were
//     loaded one keyframe earlier (DevProblem::ffr: one contiguous record per keyframe, so the prefetch needs no dependent load);
//   * the 41 sums live in registers for the whole range and are reduced once per wave; the derivative halves of the candidate (90 doubles) sit in
//     LDS, the value halves in scalar registers;
//   * every wave has the same amount of work: no tail of short blocks, the grid is one round of the machine.
// Which lane adds which block in which order is fixed by the candidate's lists, by x (the rotation below) and by W alone: runs are bitwise reproducible
// and a candidate's sums do not depend on its place in the batch. They are NOT bit-identical between launches that cut its list into a different number of
// ranges (W depends on the batch size): those agree to the rounding of a different summation order (~1e-16 of the sum of the absolute terms), like the
// sums of two frame shards. The cost tuple (association + search kernels) is untouched: bit-identical whatever the batch.
// Record j (of W) of a candidate receives the sums of its j-th range; iba_reduce2_kernel / iba_reduce_kernel add the W records in order.
#pragma once
#include "iba_kernels.hpp"

namespace iba {

constexpr int kFfrHead = 24;        // doubles in front of the relative poses of a keyframe record (DevProblem::ffr): fx fy cx cy | Tcw[12] | kp_base pt_base match_base (u64 bits) | K, n_slots (u32 pair) | 4 spare
constexpr int kFfrSlotRing = 16;    // doubles per covisible slot in the LDS ring: [R_i | t_i] (12), s t_i (3), spare
constexpr uint32_t kF2Queue = 128u; // ring capacity of a block queue: at most 63 waiting + 64 new
constexpr int kF2MaxFrames = 4095;  // keyframes a range may span (12 bits of a queue item)

struct F2Layout {   // byte offsets into the dynamic LDS of one wave (host: layout_factor2)
    uint32_t off_pre, off_cand, off_q, off_ring, off_tr, total;
    uint32_t ring_slots;   // power of two
    uint32_t ring_stride;  // doubles per ring record: kFfrHead + kFfrSlotRing * max_slots
    uint32_t ffr_stride;   // doubles per keyframe record in global memory: kFfrHead + 12 * max_slots
};

template <bool MANY, bool P2PIX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void iba_factor2_kernel(DevProblem dp, DevParams prm, const Cand* __restrict__ cands, const uint4* __restrict__ flist,
                                                                      const uint32_t* __restrict__ fcount, int flist_stride, int per_cand, double* __restrict__ partials, int nrec, int rec_base,
                                                                      int B, int W, const double* __restrict__ ffr, const double2* __restrict__ kp_c, F2Layout lay) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    uint32_t* pre = (uint32_t*)(s_raw + lay.off_pre);          // pre[f] = entries of the candidate's keyframes 0 .. f-1
    double* s_cd = (double*)(s_raw + lay.off_cand);            // dR[3][9] dt[6][3] dRlc[3][9] dtlc[6][3]
    uint2* qa = (uint2*)(s_raw + lay.off_q); uint2* qb = qa + kF2Queue; uint2* qc = qb + kF2Queue;   // plane factors / point-to-plane / point-to-point
    double* ring = (double*)(s_raw + lay.off_ring);
    double (*s_tr)[65] = (double (*)[65])(s_raw + lay.off_tr); // the final transposing reduction (aliases queues and ring: they are dead by then)
    // block -> (candidate, range): block i runs on XCD i % 8; range j of every candidate on XCD j % 8, so that the candidates' waves that walk the same
    // keyframes at the same time share one L2
    const int b = (int)((blockIdx.x >> 3) % (uint32_t)B), j = (int)(blockIdx.x & 7u) + 8 * (int)((blockIdx.x >> 3) / (uint32_t)B);
    if (j >= W) return;
    const int lane = threadIdx.x;
    const int F = dp.n_frames;
    const Cand& c = cands[b];
    // LDS hand-over between the lanes of the ONE wave of this block: the LDS executes a wave's instructions in order, so all that is needed is that the
    // compiler keeps the order — a wavefront-scope fence. (A workgroup-scope release costs s_waitcnt vmcnt(0): every prefetch of this kernel — the next
    // round's entries, the next keyframe's record — would be waited for at the next hand-over, i.e. not be a prefetch.)
    auto lds_order = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // ---- the derivative halves of the candidate -> LDS (the bodies read them as broadcast LDS loads: 90 doubles do not fit the scalar registers beside
    // R, t, Rlc, tlc — the one-wave-per-keyframe kernel spilled 144 of them to vector lanes)
    {
        const double* src0 = &c.dR[0][0]; const double* src1 = &c.dRlc[0][0];   // dR[27] dt[18] are contiguous in Cand, and so are dRlc[27] dtlc[18]
        if (lane < 45) { s_cd[lane] = src0[lane]; s_cd[45 + lane] = src1[lane]; }
    }
    const double (*s_dR)[9] = (const double (*)[9])(s_cd);
    const double (*s_dt)[3] = (const double (*)[3])(s_cd + 27);
    const double (*s_dRlc)[9] = (const double (*)[9])(s_cd + 45);
    const double (*s_dtlc)[3] = (const double (*)[3])(s_cd + 72);
    // ---- prefix sums of the candidate's list lengths
    const size_t row0 = (size_t)(per_cand ? b : 0) * (size_t)F;
    uint32_t run = 0u;
    if (lane == 0) pre[0] = 0u;
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int f = f0 + lane;
        uint32_t v = f < F ? fcount[row0 + f] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += u; }
        if (f < F) pre[f + 1] = run + v;
        run += (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    }
    const uint32_t T = run;
    // The ranges: the cut points c_j = T j / W of the candidate's list, ROTATED by an offset that depends on the candidate's place in the batch:
    // range j = [c_j + o, c_j+1 + o) of the circular list (the last range wraps: two segments). Without the rotation the waves of all the candidates
    // that share a range walk the same keyframes in lockstep and miss the L2 on the same lines at the same moment — every wave pays the full miss
    // latency on every dependent gather (measured: 237 us, 3.4 x the L2 misses of the one-wave-per-keyframe kernel, whose blocks of one keyframe
    // start at different times). Rotated, the 64 walkers of a range are spread evenly over it: each is the first to touch 1/64 of a range and
    // follows another walker's trail for the rest.
    const uint32_t c_lo = (uint32_t)(((unsigned long long)T * (unsigned)j) / (unsigned)W), c_hi = (uint32_t)(((unsigned long long)T * (unsigned)(j + 1)) / (unsigned)W);
    // (the offset is a hash of the candidate's own R, t — 64 steps of 1/64 of a range —, not its index in the batch: a candidate's sums depend on x and on
    //  the number of ranges, not on where in the batch it stands; identical candidates walk in lockstep, which costs them time only)
    uint32_t hx;
    {
        const unsigned long long u0 = (unsigned long long)__double_as_longlong(c.R[1]), u1 = (unsigned long long)__double_as_longlong(c.t[0]), u2 = (unsigned long long)__double_as_longlong(c.s);
        const uint32_t u = (uint32_t)(u0 ^ (u0 >> 32)) ^ (uint32_t)(u1 ^ (u1 >> 32)) * 0x9E3779B1u ^ (uint32_t)(u2 ^ (u2 >> 32)) * 0x85EBCA77u;
        hx = (u * 2654435761u) >> 26;   // 0 .. 63
    }
    const uint32_t rot = (uint32_t)(((unsigned long long)(T / (unsigned)W) * hx) >> 6);   // < T / W <= every range's length
    uint32_t s0_lo = 0u, s0_hi = 0u, s1_lo = 0u, s1_hi = 0u;   // up to two segments, empty ones skipped by the walk
    {
        const uint32_t lo = c_lo + rot, hi = c_hi + rot;   // (lo <= T: c_lo + rot <= c_{W-1} + T / W <= T)
        if (hi <= T) { s0_lo = lo; s0_hi = hi; }
        else { s0_lo = lo; s0_hi = T; s1_lo = 0u; s1_hi = hi - T; }
    }
    lds_order();
    NAcc A;
    for (int i = 0; i < 28; ++i) A.H[i] = 0;
    for (int i = 0; i < 7; ++i) A.b[i] = 0;
    A.chi2 = A.cost = A.nf2d = A.nfpl = A.nfpt = A.nres = 0;
    const PlaneRec* planes = prm.plane_cache ? dp.plane_local : dp.scratch_local + (size_t)(per_cand ? dp.scratch_slot_base + b : 0) * (size_t)dp.n_pt_total;
    const uint32_t rmask = lay.ring_slots - 1u, rstride = lay.ring_stride, gstride = lay.ffr_stride;
    const double cs = c.s;
    uint32_t ha = 0u, ta = 0u, hb = 0u, tb = 0u, hc = 0u, tc = 0u;   // queue heads / tails (wave-uniform)
    uint32_t f_first = 0u;   // first keyframe of the range: queue items carry their keyframe relative to it (12 bits)
    // ---- the bodies: 64 blocks of one kind, one per lane (the first `cnt` of a queue)
    auto frame_rec = [&](uint32_t item_x) -> const double* { return ring + (size_t)(((item_x >> 20) + f_first) & rmask) * rstride; };
    auto plane_batch = [&](uint32_t cnt) {
        if ((uint32_t)lane < cnt) {
            const uint2 q = qa[(ha + (uint32_t)lane) & (kF2Queue - 1u)];
            const double* fr = frame_rec(q.x);
            const uint32_t k = q.x & 0xFFFFFu;
            const unsigned long long kp_base = (unsigned long long)__double_as_longlong(fr[16]), pt_base = (unsigned long long)__double_as_longlong(fr[17]), match_base = (unsigned long long)__double_as_longlong(fr[18]);
            const unsigned long long kn = (unsigned long long)__double_as_longlong(fr[19]);
            const uint32_t K = (uint32_t)(kn & 0xFFFFFFFFull), n_sl = (uint32_t)(kn >> 32);
            const MatchPre mp = load_match_pre(dp.match_uv + match_base + k, (size_t)K, n_sl);   // with the other gathers: no load depends on another
            const float4 pt = dp.pts4[pt_base + q.y];
            const uint32_t m0 = dp.kp_fl[kp_base + k] >> 2, m1 = MANY ? (dp.kp_fl2 ? dp.kp_fl2[kp_base + k] : 0u) : 0u;
            const double p0[3] = {(double)pt.x, (double)pt.y, (double)pt.z};
            const Cam4 cam{fr[0], fr[1], fr[2], fr[3]};
            auto rel_of = [&](uint32_t sl, double* ts) { const double* rel = fr + kFfrHead + (size_t)sl * kFfrSlotRing; ts[0] = rel[12]; ts[1] = rel[13]; ts[2] = rel[14]; return rel; };
            if (P2PIX) edge_accum<MANY>(c.R, c.t, s_dR, s_dt, cam, p0, m0, m1, dp.match_uv + match_base + k, (size_t)K, mp, rel_of, prm.robust_kernel_delta, A);
            else {
                const PlaneRec& rec = planes[pt_base + q.y];
                const double n0[3] = {rec.nx, rec.ny, rec.nz};
                const double2 cz = kp_c[kp_base + k];
                plane_accum<MANY>(c.R, c.t, s_dR, s_dt, cam, cz.x, cz.y, p0, n0, m0, m1, dp.match_uv + match_base + k, (size_t)K, mp, rel_of, prm.robust_kernel_delta, A);
            }
        }
        ha += min(cnt, 64u);
#ifdef IBA_DIAG_COUNTERS
        if (lane == 0 && dp.diag) atomicAdd(dp.diag + 2, 1u);   // plane batches executed
#endif
    };
    auto p2x_batch = [&](uint2* qq, uint32_t& hh, uint32_t cnt, bool is_plane) {
        if ((uint32_t)lane < cnt) {
            const uint2 q = qq[(hh + (uint32_t)lane) & (kF2Queue - 1u)];
            const double* fr = frame_rec(q.x);
            const uint32_t k = q.x & 0xFFFFFu;
            const unsigned long long kp_base = (unsigned long long)__double_as_longlong(fr[16]), pt_base = (unsigned long long)__double_as_longlong(fr[17]);
            const float4 pt3 = dp.pts4[pt_base + q.y], mp3 = dp.kp_mp[kp_base + k];
            const double Q[3] = {(double)pt3.x, (double)pt3.y, (double)pt3.z};
            double nn[3] = {0, 0, 0};
            if (is_plane) { const PlaneRec& r3 = planes[pt_base + q.y]; nn[0] = r3.nx; nn[1] = r3.ny; nn[2] = r3.nz; }
            if (is_plane) p2pl_accum(c.Rlc, c.tlc, s_dRlc, s_dtlc, cs, fr + 4, prm.robust_kernel_3ddelta, mp3, Q, nn, A);
            else p2pt_accum(c.Rlc, c.tlc, s_dRlc, s_dtlc, cs, fr + 4, prm.robust_kernel_3ddelta, mp3, Q, A);
        }
        hh += min(cnt, 64u);
    };
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {   // (ONE copy of the walk: a rolled loop)
        const uint32_t e_lo = seg == 0 ? s0_lo : s1_lo, e_hi = seg == 0 ? s0_hi : s1_hi;
        if (e_hi <= e_lo) continue;
        // ---- first keyframe of the segment (binary search in the prefix sums): pre[f] <= e_lo < pre[f + 1]
        {
            uint32_t lo = 0u, hi = (uint32_t)F;   // invariant: pre[lo] <= e_lo < pre[hi]
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= e_lo) lo = mid; else hi = mid; }
            f_first = lo;
        }
        // ---- keyframe records: global -> registers (one keyframe ahead) -> LDS ring
        double pf0 = 0.0, pf1 = 0.0;   // elements lane, lane + 64 of the NEXT keyframe's record
        auto pf_issue = [&](uint32_t f) {
            if (f < (uint32_t)F) {
                const double* src = ffr + (size_t)f * gstride;
                if ((uint32_t)lane < gstride) pf0 = src[lane];
                if ((uint32_t)lane + 64u < gstride) pf1 = src[lane + 64];
            }
        };
        auto ring_put = [&](double* dst, uint32_t e, double v) {   // element e of a keyframe record -> its place in the ring record (+ s t_i beside a translation entry)
            if (e < (uint32_t)kFfrHead) { dst[e] = v; return; }
            const uint32_t sl = (e - (uint32_t)kFfrHead) / 12u, i = (e - (uint32_t)kFfrHead) % 12u;
            double* d = dst + kFfrHead + (size_t)sl * kFfrSlotRing;
            d[i] = v;
            if ((i & 3u) == 3u) d[12u + (i >> 2)] = v * cs;   // _t *= _s (IBACalib2.hpp:175): the product every lane of a body would form
        };
        auto stage = [&](uint32_t f) {   // the record of keyframe f (in pf0 / pf1 and, beyond 128 doubles, in global memory) -> ring slot f & rmask; then the next keyframe's loads
            double* dst = ring + (size_t)(f & rmask) * rstride;
            if ((uint32_t)lane < gstride) ring_put(dst, (uint32_t)lane, pf0);
            if ((uint32_t)lane + 64u < gstride) ring_put(dst, (uint32_t)lane + 64u, pf1);
            for (uint32_t e = (uint32_t)lane + 128u; e < gstride; e += 64u) ring_put(dst, e, ffr[(size_t)f * gstride + e]);   // (more than 8 covisible slots: the rest is fetched here)
            pf_issue(f + 1u);
        };
        pf_issue(f_first);
        stage(f_first);
        uint32_t f_hi = f_first;   // highest keyframe in the ring
        lds_order();
        // ---- the walk
        uint32_t f_l = f_first;   // this lane's keyframe
        auto locate = [&](uint32_t g, uint32_t& f) { while (g >= pre[f + 1u]) ++f; };   // (g < e_hi <= T = pre[F]: ends)
        auto entry_of = [&](uint32_t g, uint32_t f) -> uint4 {   // (read once, by this wave alone: streamed past the caches' replacement order)
            const uint4* p = flist + ((row0 + f) * (size_t)flist_stride + (g - pre[f]));
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v v = __builtin_nontemporal_load((const u4v*)p);
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        uint4 e_n = make_uint4(0u, kNone, kNone, 0u);
        uint32_t f_n = f_first;
        { const uint32_t g = e_lo + (uint32_t)lane; if (g < e_hi) { locate(g, f_n); e_n = entry_of(g, f_n); } }
        for (uint32_t cur = e_lo; cur < e_hi; cur += 64u) {
#ifdef IBA_DIAG_COUNTERS
            if (lane == 0 && dp.diag) atomicAdd(dp.diag + 3, 1u);   // rounds walked
#endif
            const uint32_t g = cur + (uint32_t)lane;
            const uint4 e = e_n;
            f_l = f_n;
            const bool live = g < e_hi;
            e_n = make_uint4(0u, kNone, kNone, 0u);
            if (g + 64u < e_hi) { locate(g + 64u, f_n); e_n = entry_of(g + 64u, f_n); }   // the next entries are in flight during this round's arithmetic
            // keyframes this round reaches: into the ring (the last live lane holds the highest). A round whose 64 entries span more keyframes than the
            // ring holds (keyframes with a handful of entries each) goes through in windows of keyframes; one window is the rule
            const uint32_t n_live = min(64u, e_hi - cur);
            const uint32_t f_max = (uint32_t)__builtin_amdgcn_readlane((int)f_l, (int)n_live - 1);
            uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f_l);
            for (;;) {
                const uint32_t w_hi = min(f_max, max(w_lo, f_hi) + lay.ring_slots - 1u);   // keyframes w_lo .. w_hi are (or will now be) in the ring together
                while (f_hi < w_hi) {
                    ++f_hi;
                    // the slot about to be overwritten belonged to keyframe f_hi - ring_slots: blocks of it (or older) that still wait in a queue go first
                    auto drain = [&](uint2* qq, uint32_t& hh, uint32_t& tt, int kind) {
                        while (tt != hh) {
                            const uint32_t fo = f_first + ((uint32_t)__builtin_amdgcn_readfirstlane((int)qq[hh & (kF2Queue - 1u)].x) >> 20);
                            if (fo + lay.ring_slots > f_hi) break;
                            if (kind == 0) plane_batch(min(tt - hh, 64u)); else p2x_batch(qq, hh, min(tt - hh, 64u), kind == 1);
                            lds_order();
                        }
                    };
                    drain(qa, ha, ta, 0); drain(qb, hb, tb, 1); drain(qc, hc, tc, 2);
                    stage(f_hi);
                    lds_order();
                }
                const uint32_t tag = (f_l - f_first) << 20;
                const bool mine = live && f_l >= w_lo && f_l <= w_hi;
                const bool hp = mine && e.y != kNone, h3 = mine && e.z != kNone, h3pl = h3 && (e.z >> 31) != 0u, h3pt = h3 && (e.z >> 31) == 0u;
                const unsigned long long bp = __ballot(hp), bpl = __ballot(h3pl), bpt = __ballot(h3pt), lt = (1ull << lane) - 1ull;
                if (hp) qa[(ta + (uint32_t)__popcll(bp & lt)) & (kF2Queue - 1u)] = make_uint2(e.x | tag, e.y);
                if (h3pl) qb[(tb + (uint32_t)__popcll(bpl & lt)) & (kF2Queue - 1u)] = make_uint2(e.x | tag, e.z & 0x7FFFFFFFu);
                if (h3pt) qc[(tc + (uint32_t)__popcll(bpt & lt)) & (kF2Queue - 1u)] = make_uint2(e.x | tag, e.z);
                ta += (uint32_t)__popcll(bp); tb += (uint32_t)__popcll(bpl); tc += (uint32_t)__popcll(bpt);
                lds_order();
                if (ta - ha >= 64u) plane_batch(64u);
                if (tb - hb >= 64u) p2x_batch(qb, hb, 64u, true);
                if (tc - hc >= 64u) p2x_batch(qc, hc, 64u, false);
                lds_order();   // the slots just read may be rewritten by the next round
                if (w_hi >= f_max) break;
                w_lo = w_hi + 1u;
            }
        }
        // end of a segment: what is left in the queues (their items carry keyframes relative to this segment's first)
        if (ta - ha) plane_batch(ta - ha);
        if (tb - hb) p2x_batch(qb, hb, tb - hb, true);
        if (tc - hc) p2x_batch(qc, hc, tc - hc, false);
        lds_order();
    }
    // ---- fixed-order reduction through LDS: every lane parks its 41 sums (two halves of <= 21 through a transposing buffer), then lane v adds the
    // 64 lanes' values of sum v in lane order
    __shared__ double s_part[48];
    double* v = (double*)&A;   // 41 contiguous doubles
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int base = half * 21, cnt = half ? 20 : 21;
#pragma unroll
        for (int q = 0; q < 21; ++q) if (q < cnt) s_tr[q][lane] = v[base + q];
        lds_order();
        if (lane < cnt) {
            double x = 0;
#pragma unroll 16
            for (int jj = 0; jj < 64; ++jj) x += s_tr[lane][jj];
            s_part[base + lane] = x;
        }
        lds_order();
    }
    {
        const int i = lane;
        double out = 0;
        int src = -1;   // NAcc slot -> partial slot
        if (i >= P_H0 && i < P_H0 + 28) src = i - P_H0;
        else if (i >= P_B0 && i < P_B0 + 7) src = 28 + (i - P_B0);
        else if (i == P_CHI2) src = 35; else if (i == P_COST) src = 36; else if (i == P_NF_3D2D) src = 37;
        else if (i == P_NF_P2PL) src = 38; else if (i == P_NF_P2PT) src = 39; else if (i == P_NRES) src = 40;
        if (src >= 0) out = s_part[src];
        partials[((size_t)b * nrec + rec_base + j) * kPartialStride + i] = out;
    }
}

}  // namespace iba
