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
                                                                      int B, int W, const double* __restrict__ ffr, const double2* __restrict__ kp_c, F2Layout lay, int dbg) {
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
    for (int f0 = 0; f0 < F; f0 += 256) {   // four consecutive keyframes per lane: the four loads in flight together, ONE wave scan per 256 keyframes
        const int f = f0 + 4 * lane;
        uint32_t c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = f + q < F ? fcount[row0 + f + q] : 0u;
        c4[1] += c4[0]; c4[2] += c4[1]; c4[3] += c4[2];
        uint32_t v = c4[3];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += u; }
        const uint32_t before = run + v - c4[3];   // entries of this chunk in front of this lane's four keyframes
#pragma unroll
        for (int q = 0; q < 4; ++q) if (f + q < F) pre[f + q + 1] = before + c4[q];
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
    // (the offset is a hash of the candidate's own R, t, s — 1024 steps of a range's length —, not its index in the batch: a candidate's sums depend on x and on
    //  the number of ranges, not on where in the batch it stands; identical candidates walk in lockstep, which costs them time only)
    uint32_t hx;
    {
        const unsigned long long u0 = (unsigned long long)__double_as_longlong(c.R[1]), u1 = (unsigned long long)__double_as_longlong(c.t[0]), u2 = (unsigned long long)__double_as_longlong(c.s);
        const uint32_t u = (uint32_t)(u0 ^ (u0 >> 32)) ^ (uint32_t)(u1 ^ (u1 >> 32)) * 0x9E3779B1u ^ (uint32_t)(u2 ^ (u2 >> 32)) * 0x85EBCA77u;
        hx = (u * 2654435761u) >> 22;   // 0 .. 1023
    }
    const uint32_t rot = (uint32_t)(((unsigned long long)(T / (unsigned)W) * hx) >> 10);   // < T / W <= every range's length
    uint32_t s0_lo = 0u, s0_hi = 0u, s1_lo = 0u, s1_hi = 0u;   // up to two segments, empty ones skipped by the walk
    {
        const uint32_t lo = c_lo + rot, hi = c_hi + rot;   // (lo <= T: c_lo + rot <= c_{W-1} + T / W <= T)
        if (hi <= T) { s0_lo = lo; s0_hi = hi; }
        else { s0_lo = lo; s0_hi = T; s1_lo = 0u; s1_hi = hi - T; }
    }
    lds_order();
    NAccP A;
    for (int i = 0; i < 28; ++i) A.H[i] = 0;
    for (int i = 0; i < 7; ++i) A.b[i] = 0;
    A.chi2 = A.cost = 0;
    A.c2d_pl = A.cpt = A.rows = 0u;
    const PlaneRec* planes = prm.plane_cache ? dp.plane_local : dp.scratch_local + (size_t)(per_cand ? dp.scratch_slot_base + b : 0) * (size_t)dp.n_pt_total;
    const uint32_t rmask = lay.ring_slots - 1u, rstride = lay.ring_stride, gstride = lay.ffr_stride;
    const double cs = c.s;
    uint32_t ha = 0u, ta = 0u, hb = 0u, tb = 0u, hc = 0u, tc = 0u;   // queue heads / tails (wave-uniform)
    uint32_t f_first = 0u;   // first keyframe of the range: queue items carry their keyframe relative to it (12 bits)
    uint32_t f_ring_any = 0u;   // a keyframe whose record is in the ring right now (the newest): where lanes without a block point their gathers
    // ---- the bodies: 64 blocks of one kind, one per lane (the first `cnt` of a queue)
    auto frame_rec = [&](uint32_t item_x) -> const double* { return ring + (size_t)(((item_x >> 20) + f_first) & rmask) * rstride; };
    // (dbg: timing cuts — results invalid. bit 0: the plane-factor bodies are skipped, bit 1: the 3d-3d bodies)
    // A body is two halves: its GATHERS (queue item -> keyframe record -> scan point, plane normal, keypoint ray, flag word, matches: no load depends on
    // another) and its ARITHMETIC. The walk issues the gathers of a plane-factor batch and of a point-to-plane batch together and then runs the two
    // arithmetic halves: one exposed round trip for both instead of one each (their queues fill at the same pace: most entries carry both blocks).
    struct PIn { bool on; uint32_t fro, mro; uint32_t m0, m1, K; float px, py, pz; double n0x, n0y, n0z, czx, czy; MatchPre mp; };   // fro: the keyframe record's place in the ring, mro: the keypoint's row in the match table (32-bit offsets: one register each)
    struct QIn { bool on; uint32_t fro; float qx, qy, qz, mx, my, mz; double nx, ny, nz; };
    // (the gathers are UNCONDITIONAL: a lane without a block reads element 0 of every table. A load under `if (lane has a block)` whose result is merged
    //  with a default afterwards is waited for at the merge — right behind its issue —, which is exactly what the split into two halves is there to avoid)
    auto plane_load = [&](uint32_t cnt) -> PIn {
        PIn in;
        in.on = (uint32_t)lane < cnt && !(dbg & 1);
        uint2 q = make_uint2(0u, 0u);
        if (in.on) q = qa[(ha + (uint32_t)lane) & (kF2Queue - 1u)];
        const double* fr = in.on ? frame_rec(q.x) : ring + (size_t)(f_ring_any & rmask) * rstride;   // (any keyframe record that is in the ring)
        const uint32_t k = q.x & 0xFFFFFu;
        const unsigned long long kp_base = (unsigned long long)__double_as_longlong(fr[16]), pt_base = (unsigned long long)__double_as_longlong(fr[17]), match_base = (unsigned long long)__double_as_longlong(fr[18]);
        const unsigned long long kn = (unsigned long long)__double_as_longlong(fr[19]);
        const uint32_t K = (uint32_t)(kn & 0xFFFFFFFFull), n_sl = (uint32_t)(kn >> 32);
        size_t kp_i = in.on ? (size_t)(kp_base + k) : 0, pt_i = in.on ? (size_t)(pt_base + q.y) : 0;
        if (dbg & 16) { kp_i = (size_t)lane; pt_i = (size_t)lane; }   // (timing cut: every gather of the launch in the same few cache lines — what the kernel costs without its L2 traffic)
        const bool has_m = in.on && n_sl > 0u && !(dbg & 16);
        in.fro = (uint32_t)(fr - ring); in.K = K; in.mro = has_m ? (uint32_t)(match_base + k) : 0u;
        {   // the matches of the first kMatchPre slots (a keyframe with fewer: its last slot again; the bits of the flag word say which count)
            const float2* mrow = dp.match_uv + in.mro;
            const size_t s1 = has_m ? (size_t)min(1u, n_sl - 1u) * K : 0, s2 = has_m ? (size_t)min(2u, n_sl - 1u) * K : 0;
            const float2 a = mrow[0], bq = mrow[s1], cq = mrow[s2];
            in.mp = MatchPre{a.x, a.y, bq.x, bq.y, cq.x, cq.y};
        }
        const float4 pt = dp.pts4[pt_i];
        in.px = pt.x; in.py = pt.y; in.pz = pt.z;
        in.m0 = dp.kp_fl[kp_i] >> 2; in.m1 = MANY ? (dp.kp_fl2 ? dp.kp_fl2[kp_i] : 0u) : 0u;
        if (!P2PIX) {
            const PlaneRec& rec = planes[pt_i];
            in.n0x = rec.nx; in.n0y = rec.ny; in.n0z = rec.nz;
            const double2 cz = kp_c[kp_i];
            in.czx = cz.x; in.czy = cz.y;
        } else { in.n0x = in.n0y = in.n0z = in.czx = in.czy = 0.0; }
        ha += min(cnt, 64u);
        return in;
    };
    auto plane_compute = [&](const PIn& in) {
        if (in.on && (dbg & 4)) { A.cost += (double)in.px + in.n0x + in.czx + (double)in.mp.u0 + (double)in.mp.u1 + (double)in.mp.u2 + (double)in.m0; return; }   // (timing cut: the gathers without the arithmetic)
        if (in.on) {
            const double* fr = ring + in.fro;
            const float2* mrow = dp.match_uv + in.mro;
            const double p0[3] = {(double)in.px, (double)in.py, (double)in.pz};
            const Cam4 cam{fr[0], fr[1], fr[2], fr[3]};
            auto rel_of = [&](uint32_t sl, double* ts) { const double* rel = fr + kFfrHead + (size_t)sl * kFfrSlotRing; ts[0] = rel[12]; ts[1] = rel[13]; ts[2] = rel[14]; return rel; };
            if (P2PIX) edge_accum<MANY>(c.R, c.t, s_dR, s_dt, cam, p0, in.m0, in.m1, mrow, (size_t)in.K, in.mp, rel_of, prm.robust_kernel_delta, A);
            else {
                const double n0[3] = {in.n0x, in.n0y, in.n0z};
                plane_accum<MANY>(c.R, c.t, s_dR, s_dt, cam, in.czx, in.czy, p0, n0, in.m0, in.m1, mrow, (size_t)in.K, in.mp, rel_of, prm.robust_kernel_delta, A);
            }
        }
    };
    auto p2x_load = [&](uint2* qq, uint32_t& hh, uint32_t cnt, bool is_plane) -> QIn {
        QIn in;
        in.on = (uint32_t)lane < cnt && !(dbg & 2);
        uint2 q = make_uint2(0u, 0u);
        if (in.on) q = qq[(hh + (uint32_t)lane) & (kF2Queue - 1u)];
        const double* fr = in.on ? frame_rec(q.x) : ring + (size_t)(f_ring_any & rmask) * rstride;
        const uint32_t k = q.x & 0xFFFFFu;
        const unsigned long long kp_base = (unsigned long long)__double_as_longlong(fr[16]), pt_base = (unsigned long long)__double_as_longlong(fr[17]);
        size_t kp_i = in.on ? (size_t)(kp_base + k) : 0, pt_i = in.on ? (size_t)(pt_base + q.y) : 0;
        if (dbg & 16) { kp_i = (size_t)lane; pt_i = (size_t)lane; }
        const float4 pt3 = dp.pts4[pt_i], mp3 = dp.kp_mp[kp_i];
        in.fro = (uint32_t)(fr - ring); in.qx = pt3.x; in.qy = pt3.y; in.qz = pt3.z; in.mx = mp3.x; in.my = mp3.y; in.mz = mp3.z;
        if (is_plane) { const PlaneRec& r3 = planes[pt_i]; in.nx = r3.nx; in.ny = r3.ny; in.nz = r3.nz; } else { in.nx = in.ny = in.nz = 0.0; }
        hh += min(cnt, 64u);
        return in;
    };
    auto p2x_compute = [&](const QIn& in, bool is_plane) {
        if (in.on && (dbg & 8)) { A.cost += (double)in.qx + (double)in.mx + in.nx; return; }   // (timing cut)
        if (in.on) {
            const double Q[3] = {(double)in.qx, (double)in.qy, (double)in.qz};
            const float4 mp3 = make_float4(in.mx, in.my, in.mz, 0.f);
            if (is_plane) { const double nn[3] = {in.nx, in.ny, in.nz}; p2pl_accum(c.Rlc, c.tlc, s_dRlc, s_dtlc, cs, ring + in.fro + 4, prm.robust_kernel_3ddelta, mp3, Q, nn, A); }
            else p2pt_accum(c.Rlc, c.tlc, s_dRlc, s_dtlc, cs, ring + in.fro + 4, prm.robust_kernel_3ddelta, mp3, Q, A);
        }
    };
    auto plane_batch = [&](uint32_t cnt) { const PIn in = plane_load(cnt); plane_compute(in); };
    auto p2x_batch = [&](uint2* qq, uint32_t& hh, uint32_t cnt, bool is_plane) { const QIn in = p2x_load(qq, hh, cnt, is_plane); p2x_compute(in, is_plane); };
#ifdef IBA_DIAG_COUNTERS
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, t_begin = __builtin_readcyclecounter();   // cycles per phase of the walk (see the end of the kernel)
#define F2_TICK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); tph[i] += _t - tlast; tlast = _t; } while (0)
#else
#define F2_TICK(i) do { } while (0)
#endif
    QIn ib; ib.on = false; ib.fro = 0u; ib.qx = ib.qy = ib.qz = ib.mx = ib.my = ib.mz = 0.f; ib.nx = ib.ny = ib.nz = 0.0;   // the point-to-plane batch whose gathers are in flight (evaluated one round later)
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
        // ---- keyframe records: global -> LDS ring, up to four keyframes at a time, fetched when the walk first needs one of them. (Round 6, first version:
        // one keyframe ahead in a register across the rounds. The compiler cannot count the loads issued since an earlier iteration and waits with
        // vmcnt(0) — behind the entries just requested: a full memory latency per keyframe, 20 % of the kernel. Four at a time, waited for on the spot,
        // cost one latency per four keyframes, and at the bench shape a range's seven keyframes are in the ring after the prologue and one more fetch.)
        auto ring_put = [&](double* dst, uint32_t e, double v) {   // element e of a keyframe record -> its place in the ring record (+ s t_i beside a translation entry)
            if (e < (uint32_t)kFfrHead) { dst[e] = v; return; }
            const uint32_t sl = (e - (uint32_t)kFfrHead) / 12u, i = (e - (uint32_t)kFfrHead) % 12u;
            double* d = dst + kFfrHead + (size_t)sl * kFfrSlotRing;
            d[i] = v;
            if ((i & 3u) == 3u) d[12u + (i >> 2)] = v * cs;   // _t *= _s (IBACalib2.hpp:175): the product every lane of a body would form
        };
        auto stage_range = [&](uint32_t f_from, uint32_t f_to) {   // the records of the keyframes f_from .. f_to (at most four) -> their ring slots
            const uint32_t n = f_to - f_from + 1u;
            double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            if ((uint32_t)lane < gstride) {
                const double* src = ffr + (size_t)f_from * gstride + lane;
                v0 = src[0];
                if (n > 1u) v1 = src[gstride];
                if (n > 2u) v2 = src[2 * (size_t)gstride];
                if (n > 3u) v3 = src[3 * (size_t)gstride];
                ring_put(ring + (size_t)(f_from & rmask) * rstride, (uint32_t)lane, v0);
                if (n > 1u) ring_put(ring + (size_t)((f_from + 1u) & rmask) * rstride, (uint32_t)lane, v1);
                if (n > 2u) ring_put(ring + (size_t)((f_from + 2u) & rmask) * rstride, (uint32_t)lane, v2);
                if (n > 3u) ring_put(ring + (size_t)((f_from + 3u) & rmask) * rstride, (uint32_t)lane, v3);
            }
            for (uint32_t f = f_from; f <= f_to; ++f)   // (more than three covisible slots: the rest of a record)
                for (uint32_t e = (uint32_t)lane + 64u; e < gstride; e += 64u) ring_put(ring + (size_t)(f & rmask) * rstride, e, ffr[(size_t)f * gstride + e]);
        };
        const uint32_t stage_extra = lay.ring_slots / 2u - 1u;   // keyframes fetched beyond the one the walk asks for
        stage_range(f_first, min((uint32_t)F - 1u, f_first + min(3u, lay.ring_slots - 1u)));
        uint32_t f_hi = min((uint32_t)F - 1u, f_first + min(3u, lay.ring_slots - 1u));   // highest keyframe in the ring
        f_ring_any = f_first;
        lds_order();
        // ---- the walk
        uint32_t f_l = f_first;   // this lane's keyframe
        auto locate = [&](uint32_t g, uint32_t& f) { while (g >= pre[f + 1u]) ++f; };   // (g < e_hi <= T = pre[F]: ends)
        auto entry_of = [&](uint32_t g, uint32_t f) -> uint4 {   // (read once, by this wave alone: streamed past the caches' replacement order)
            const uint4* p = flist + ((row0 + f) * (size_t)flist_stride + (g - pre[f]));
            // (three of the entry's four words: the flag word is not read here, and a destination register that is dead on arrival is handed out again
            //  at once — the write to it then waits for the load, vmcnt(0) right behind the issue: seen in the ISA)
            typedef uint32_t u3v __attribute__((ext_vector_type(3)));
            const u3v v = __builtin_nontemporal_load((const u3v*)p);
            return make_uint4(v.x, v.y, v.z, 0u);
        };
        // (the entry loads are UNCONDITIONAL — lanes past the end of the segment read its last entry again and are switched off by `live`: a load under
        //  `if (in range)` merged with a default is waited for at the merge, i.e. at once, and the prefetch of the next round's entries was none)
        uint32_t f_n = f_first;
        uint4 e_n;
        { const uint32_t g = min(e_lo + (uint32_t)lane, e_hi - 1u); locate(g, f_n); e_n = entry_of(g, f_n); }
        for (uint32_t cur = e_lo; cur < e_hi; cur += 64u) {
#ifdef IBA_DIAG_COUNTERS
            tlast = __builtin_readcyclecounter();
#endif
            const uint32_t g = cur + (uint32_t)lane;
            const uint4 e = e_n;
            f_l = f_n;
            const bool live = g < e_hi;
            { const uint32_t gn = min(g + 64u, e_hi - 1u); locate(gn, f_n); e_n = entry_of(gn, f_n); }   // the next entries are in flight during this round's arithmetic
            // keyframes this round reaches: into the ring (the last live lane holds the highest). A round whose 64 entries span more keyframes than the
            // ring holds (keyframes with a handful of entries each) goes through in windows of keyframes; one window is the rule
#ifdef IBA_DIAG_COUNTERS
            asm volatile("" :: "v"(e.x));
#endif
            F2_TICK(0);   // this round's entries have landed, the next round's are issued
            const uint32_t n_live = min(64u, e_hi - cur);
            const uint32_t f_max = (uint32_t)__builtin_amdgcn_readlane((int)f_l, (int)n_live - 1);
            uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)f_l);
            for (;;) {
                const uint32_t w_hi = min(f_max, w_lo + lay.ring_slots - 1u);   // keyframes w_lo .. w_hi are (or will now be) in the ring together (f_hi <= w_lo + ring_slots - 1 always: see f_to)
                while (f_hi < w_hi) {
                    // the keyframes the window needs and a few beyond, as far as the ring allows: every keyframe of the window [w_lo, w_hi] stays in it
                    const uint32_t f_to = min(min(f_hi + 4u, (uint32_t)F - 1u), max(w_hi, min(w_hi + stage_extra, w_lo + lay.ring_slots - 1u)));
                    if (__builtin_amdgcn_readfirstlane((int)__ballot(ib.on) != 0 ? 1 : 0)) { p2x_compute(ib, true); ib.on = false; }   // (the gathered batch reads its keyframes' poses from the ring at evaluation time)
                    // the slots about to be overwritten belonged to the keyframes <= f_to - ring_slots: blocks of those that still wait in a queue go first
                    auto drain = [&](uint2* qq, uint32_t& hh, uint32_t& tt, int kind) {
                        while (tt != hh) {
                            const uint32_t fo = f_first + ((uint32_t)__builtin_amdgcn_readfirstlane((int)qq[hh & (kF2Queue - 1u)].x) >> 20);
                            if (fo + lay.ring_slots > f_to) break;
                            if (kind == 0) plane_batch(min(tt - hh, 64u)); else p2x_batch(qq, hh, min(tt - hh, 64u), kind == 1);
                            lds_order();
                        }
                    };
                    drain(qa, ha, ta, 0); drain(qb, hb, tb, 1); drain(qc, hc, tc, 2);
                    stage_range(f_hi + 1u, f_to);
                    f_hi = f_to;
                    f_ring_any = f_hi;
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
                F2_TICK(1);   // keyframes staged, blocks queued
                {   // the gathers of both bodies that are due are issued together, then the two arithmetic halves run (the lighter first: the plane factors'
                    // inputs wait in registers beside it, not the other way round). Evaluating the point-to-plane batch one round LATER — its gathers
                    // in flight across the round — was measured: 0.130 -> 0.138 ms. A wait for the next round's entries cannot be told apart from a wait
                    // for everything issued since (loads return in order, and the compiler's count is lost across the loop), so the latency moved, not went.
                    const PIn ia = plane_load(ta - ha >= 64u ? 64u : 0u);
                    ib = p2x_load(qb, hb, tb - hb >= 64u ? 64u : 0u, true);
                    F2_TICK(2);   // gathers issued
                    p2x_compute(ib, true); ib.on = false;
#ifdef IBA_DIAG_COUNTERS
                    asm volatile("" :: "v"(A.H[0]));
#endif
                    F2_TICK(3);   // point-to-plane arithmetic (incl. the wait for its gathers)
                    plane_compute(ia);
#ifdef IBA_DIAG_COUNTERS
                    asm volatile("" :: "v"(A.H[27]));
#endif
                    F2_TICK(4);   // plane-factor arithmetic
                }
                if (tc - hc >= 64u) p2x_batch(qc, hc, 64u, false);
                lds_order();   // the slots just read may be rewritten by the next round
                F2_TICK(5);   // point-to-point batch
                if (w_hi >= f_max) break;
                w_lo = w_hi + 1u;
            }
        }
        // end of a segment: the gathered batch, then what is left in the queues (their items carry keyframes relative to this segment's first)
        p2x_compute(ib, true); ib.on = false;
        if (ta - ha) plane_batch(ta - ha);
        if (tb - hb) p2x_batch(qb, hb, tb - hb, true);
        if (tc - hc) p2x_batch(qc, hc, tc - hc, false);
        lds_order();
    }
    // ---- fixed-order reduction through LDS: every lane parks its 37 sums (two halves of <= 21 through a transposing buffer), then lane v adds the
    // 64 lanes' values of sum v in lane order; the four counters are integers: a wave-wide add
    __shared__ double s_part[48];
    double* v = (double*)&A;   // 37 contiguous doubles: H[28] b[7] chi2 cost
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int base = half * 21, cnt = half ? 16 : 21;
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
    uint32_t n2d = A.c2d_pl & 0xFFFFu, npl = A.c2d_pl >> 16, npt = A.cpt, nrows = A.rows;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n2d += (uint32_t)__shfl_xor((int)n2d, d); npl += (uint32_t)__shfl_xor((int)npl, d); npt += (uint32_t)__shfl_xor((int)npt, d); nrows += (uint32_t)__shfl_xor((int)nrows, d); }
    {
        const int i = lane;
        double out = 0;
        int src = -1;   // accumulator slot -> partial slot
        if (i >= P_H0 && i < P_H0 + 28) src = i - P_H0;
        else if (i >= P_B0 && i < P_B0 + 7) src = 28 + (i - P_B0);
        else if (i == P_CHI2) src = 35; else if (i == P_COST) src = 36;
        if (src >= 0) out = s_part[src];
#ifdef IBA_DIAG_COUNTERS
        if (i >= 56 && i < 62) out = (double)tph[i - 56];
        if (i == 62) out = (double)(__builtin_readcyclecounter() - t_begin);   // the wave's whole life
        if (i == 63) out = 1.0;                                                // waves
#endif
        if (i == P_NF_3D2D) out = (double)n2d; else if (i == P_NF_P2PL) out = (double)npl; else if (i == P_NF_P2PT) out = (double)npt; else if (i == P_NRES) out = (double)nrows;
        partials[((size_t)b * nrec + rec_base + j) * kPartialStride + i] = out;
    }
}

}  // namespace iba
