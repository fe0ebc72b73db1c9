// Host-side caller of the cost path: a batch-aware mesh adaptive direct search for the global stage
// (SURVEY.md 8(f) row 2). The reference hands BALoss::eval_x to NOMAD 4 (third party, absent here; set-up at
// iba_global.cpp:551-591): 7 variables, box bounds around the hand-eye initialiser, one objective and three
// progressive-barrier constraints (OBJ PB PB PB), OrthoMADS 2N poll directions, initial frame size 0.5 per variable,
// minimum mesh size 1e-6, 5000 black-box evaluations, evaluated one point at a time on one thread.
//
// This is not a re-implementation of NOMAD (its search steps, quadratic models and VNS are not restated). It is the
// published MADS skeleton it is built on — Audet & Dennis 2006 (mesh / frame sizes 4^-l / 2^-l), Abramson et al. 2009
// (OrthoMADS: Halton direction -> Householder basis -> 2n mesh directions), Audet & Dennis 2009 (progressive barrier
// with h = sum max(c_j, 0)^2, feasible + infeasible incumbents, h_max update) — arranged for a black box that costs
// the same for 1 and for 64 points: every iteration evaluates ONE batch = full polls (several orthogonal bases)
// around the feasible and the infeasible incumbent plus a speculative point along the last successful direction.
// Opportunism inside a poll is meaningless here and is dropped; everything is deterministic for a given seed.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

namespace iba {

constexpr int kMadsN = 7;

struct MadsOptions {
    int max_bb_eval = 5000;                         // max_bbeval (iba_calib_global.yml:42)
    double lb[kMadsN], ub[kMadsN];                  // absolute bounds (x0 + lb, x0 + ub in the reference, iba_global.cpp:530-533)
    double init_frame[kMadsN];                      // INITIAL_POLL_SIZE (init_frame: 0.5 each)
    double min_mesh = 1e-6;                         // MIN_MESH_SIZE
    int seed = 0;
    int bases_per_poll = 2;                         // orthogonal 2n-bases per poll centre and iteration (1 = plain OrthoMADS 2N)
    int max_batch = 64;                             // points per black-box call
    bool speculative = true;
    double frame_box_fraction = 0.1;                // first frame = min(init_frame, fraction * (ub - lb)): a frame wider than
                                                    // the box only ever polls the box faces
    // variable-neighbourhood restarts once a descent has reached the minimum mesh (the role of NOMAD's VNS_MADS_SEARCH,
    // use_vns: true in iba_calib_global.yml:45)
    int vns_max_idle = 6;                           // stop after this many consecutive restarts without improvement (0: no restarts)
    int vns_max_k = 6;                              // largest shake, in multiples of vns_amplitude * first frame
    double vns_amplitude = 0.5;
    int vns_mesh_index = 0;                         // mesh index a restarted descent begins with
    std::vector<double>* trace = nullptr;           // optional: every point handed to the black box, in order, as 8 doubles (x, f)
    std::vector<int>* batch_sizes = nullptr;        // optional: the size of every black-box call, in order (the rows of `trace` split into calls)
};
struct MadsPoint { double x[kMadsN]; double f, c[3], h; };
struct MadsResult {
    MadsPoint best;
    int feasible = 0, evaluations = 0, iterations = 0, batches = 0, cache_hits = 0;
    int stop_reason = 0;                            // 1 = min mesh reached, 2 = evaluation budget, 3 = black box failed
    int mesh_index = 0, restarts = 0;
};

inline double mads_h(const double* c) {
    double h = 0;
    for (int j = 0; j < 3; ++j) { if (!(c[j] <= 0)) { if (c[j] != c[j]) return std::numeric_limits<double>::infinity(); h += c[j] * c[j]; } }
    return h;
}
// t-th element of the Halton sequence in base p
inline double mads_halton(unsigned t, unsigned p) {
    double f = 1.0, r = 0.0;
    while (t > 0) { f /= p; r += f * (t % p); t /= p; }
    return r;
}

// EvalBatch(const double* X /*B x 7*/, int B, MadsPoint* out) -> bool : fills f and c[3] of every point
template <class EvalBatch>
inline bool mads_minimize(const double* x0, const MadsOptions& o, EvalBatch eval, MadsResult& res) {
    constexpr int n = kMadsN;
    static const unsigned primes[n] = {2, 3, 5, 7, 11, 13, 17};
    res = MadsResult();
    const double inf = std::numeric_limits<double>::infinity();
    std::map<std::array<long long, n>, MadsPoint> cache;
    auto key_of = [&](const double* x) {
        std::array<long long, n> k;
        for (int i = 0; i < n; ++i) k[i] = (long long)std::llround(x[i] / (o.min_mesh * 0.0625));
        return k;
    };
    auto clamp = [&](double* x) { for (int i = 0; i < n; ++i) x[i] = std::min(std::max(x[i], o.lb[i]), o.ub[i]); };
    double frame0[n];
    for (int i = 0; i < n; ++i) {
        const double box = o.frame_box_fraction * (o.ub[i] - o.lb[i]);
        frame0[i] = (box > 0) ? std::min(o.init_frame[i], box) : o.init_frame[i];
    }
    unsigned halton_t = 7u + (unsigned)std::abs(o.seed) * 101u;
    bool failed = false;

    // evaluate a list of trial points (cached ones are not re-evaluated), in chunks of max_batch
    auto run = [&](std::vector<MadsPoint>& trial) -> bool {
        std::vector<int> todo;
        std::vector<std::array<long long, n>> keys(trial.size());   // (each key once: the duplicate test below compared freshly rounded keys pairwise — 10 k llround per batch of 38, half of the driver's own time)
        for (size_t i = 0; i < trial.size(); ++i) {
            trial[i].f = inf; trial[i].h = inf;   // stays rejected unless the cache or the black box fills it in
            keys[i] = key_of(trial[i].x);
            auto it = cache.find(keys[i]);
            if (it != cache.end()) { trial[i] = it->second; ++res.cache_hits; continue; }
            bool dup = false;
            for (int j : todo) if (keys[(size_t)j] == keys[i]) { dup = true; break; }
            if (!dup) todo.push_back((int)i);
        }
        const int budget = o.max_bb_eval - res.evaluations;
        if ((int)todo.size() > budget) todo.resize(std::max(budget, 0));
        for (size_t s = 0; s < todo.size(); s += (size_t)o.max_batch) {
            const int B = (int)std::min<size_t>((size_t)o.max_batch, todo.size() - s);
            std::vector<double> X((size_t)B * n);
            std::vector<MadsPoint> out(B);
            for (int b = 0; b < B; ++b) std::memcpy(&X[(size_t)b * n], trial[todo[s + b]].x, sizeof(double) * n);
            if (!eval(X.data(), B, out.data())) return false;
            ++res.batches;
            if (o.batch_sizes) o.batch_sizes->push_back(B);
            for (int b = 0; b < B; ++b) {
                MadsPoint& p = trial[todo[s + b]];
                p.f = out[b].f; std::memcpy(p.c, out[b].c, sizeof(p.c));
                p.h = (p.f == p.f) ? mads_h(p.c) : inf;
                cache[keys[(size_t)todo[s + b]]] = p;
                ++res.evaluations;
                if (o.trace) { o.trace->insert(o.trace->end(), p.x, p.x + n); o.trace->push_back(p.f); }
            }
        }
        return true;
    };
    auto halton_dir = [&](double* q) {   // unit vector from the next Halton point
        double nq = 0;
        for (int i = 0; i < n; ++i) { q[i] = 2.0 * mads_halton(halton_t, primes[i]) - 1.0; nq += q[i] * q[i]; }
        ++halton_t;
        nq = std::sqrt(nq);
        if (!(nq > 1e-12)) { for (int i = 0; i < n; ++i) q[i] = 0; q[0] = 1.0; nq = 1.0; }
        for (int i = 0; i < n; ++i) q[i] /= nq;
    };

    // One MADS descent with the progressive barrier, from `start` and mesh index l0 down to the minimum mesh size.
    // Returns its best point in `best` (feasible if it saw any feasible point).
    auto descent = [&](const double* start, int l0, MadsPoint& best, bool& best_feasible) -> bool {
        bool haveF = false, haveI = false;
        MadsPoint xF{}, xI{};
        double hmax = inf;
        std::vector<MadsPoint> seen;   // points of this descent (for the h_max update)
        auto absorb = [&](const std::vector<MadsPoint>& pts) {   // 2 = dominating success, 1 = improving, 0 = none
            int success = 0;
            for (const MadsPoint& p : pts) {
                if (!(p.h < inf) || !(p.f == p.f)) continue;
                seen.push_back(p);
                if (p.h == 0.0) {
                    if (!haveF || p.f < xF.f) { xF = p; haveF = true; success = 2; }
                } else if (p.h <= hmax) {
                    if (!haveI) { xI = p; haveI = true; success = std::max(success, 1); }
                    else if ((p.h < xI.h && p.f <= xI.f) || (p.h <= xI.h && p.f < xI.f)) { xI = p; success = 2; }
                    else if (p.h < xI.h) success = std::max(success, 1);
                }
            }
            return success;
        };
        MadsPoint c0{};
        std::memcpy(c0.x, start, sizeof(double) * n);
        clamp(c0.x);
        {
            std::vector<MadsPoint> t(1, c0);
            if (!run(t)) return false;
            absorb(t);
        }
        int l = l0;
        double last_dir[n] = {0};
        bool have_dir = false;
        for (;;) {
            double frame[n], mesh[n];
            bool fine = true;
            for (int i = 0; i < n; ++i) {
                frame[i] = frame0[i] * std::ldexp(1.0, -l);
                mesh[i] = frame0[i] * std::ldexp(1.0, -2 * l);
                if (mesh[i] >= o.min_mesh) fine = false;
            }
            res.mesh_index = l;
            if (fine || res.evaluations >= o.max_bb_eval) break;
            ++res.iterations;
            std::vector<MadsPoint> trial;
            auto add_poll = [&](const MadsPoint& ctr) {
                for (int bidx = 0; bidx < o.bases_per_poll; ++bidx) {
                    // OrthoMADS: Halton point -> unit vector q -> Householder H = I - 2 q q^T (orthogonal columns)
                    double q[n];
                    halton_dir(q);
                    for (int j = 0; j < n; ++j) {
                        double col[n], cmax = 0;
                        for (int i = 0; i < n; ++i) { col[i] = (i == j ? 1.0 : 0.0) - 2.0 * q[i] * q[j]; cmax = std::max(cmax, std::fabs(col[i])); }
                        for (int sgn = 0; sgn < 2; ++sgn) {
                            MadsPoint p{};
                            for (int i = 0; i < n; ++i) {
                                const double step = (sgn ? -1.0 : 1.0) * frame[i] * col[i] / cmax;   // |.|_inf = frame size
                                p.x[i] = ctr.x[i] + mesh[i] * std::nearbyint(step / mesh[i]);         // on the mesh
                            }
                            clamp(p.x);
                            trial.push_back(p);
                        }
                    }
                }
            };
            if (haveF) add_poll(xF);
            if (haveI) add_poll(xI);
            if (!haveF && !haveI) add_poll(c0);   // the start was rejected by the barrier (h = inf): poll around it anyway
            if (o.speculative && have_dir) {       // one step further along the direction that just succeeded
                const MadsPoint& ctr = haveF ? xF : xI;
                MadsPoint p{};
                for (int i = 0; i < n; ++i) p.x[i] = ctr.x[i] + mesh[i] * std::nearbyint(last_dir[i] / mesh[i]);
                clamp(p.x);
                trial.push_back(p);
            }
            const MadsPoint oldF = xF, oldI = xI;
            const bool hadF = haveF, hadI = haveI;
            if (!run(trial)) return false;
            const int success = absorb(trial);
            have_dir = false;
            if (success == 2) {
                const bool viaF = haveF && (!hadF || xF.f < oldF.f);
                const MadsPoint& now = viaF ? xF : xI;
                const MadsPoint& was = viaF ? (hadF ? oldF : now) : (hadI ? oldI : now);
                double nd = 0;
                for (int i = 0; i < n; ++i) { last_dir[i] = now.x[i] - was.x[i]; nd += last_dir[i] * last_dir[i]; }
                have_dir = nd > 0;
                l = std::max(l - 1, 0);
            } else if (success == 1) {
                // improving: keep the mesh, tighten the barrier to the largest h below the infeasible incumbent's
                double hm = 0;
                for (const MadsPoint& p : seen) if (p.h < xI.h && p.h > hm) hm = p.h;
                hmax = hm > 0 ? hm : xI.h;
                if (haveI && xI.h > hmax) {   // the best infeasible point still under the barrier
                    haveI = false;
                    for (const MadsPoint& p : seen) if (p.h > 0 && p.h <= hmax && (!haveI || p.f < xI.f)) { xI = p; haveI = true; }
                }
            } else {
                if (haveI) hmax = xI.h;
                ++l;
            }
        }
        if (haveF) { best = xF; best_feasible = true; }
        else if (haveI) { best = xI; best_feasible = false; }
        else {
            best = c0; best.f = inf; best.h = inf; best_feasible = false;
            for (const MadsPoint& p : seen) if (p.h < best.h) best = p;
        }
        return true;
    };
    auto better = [&](const MadsPoint& a, bool fa, const MadsPoint& b, bool fb) {
        if (fa != fb) return fa;
        // a restart has to beat the incumbent by more than round-off noise, or two equivalent answers would keep
        // resetting the idle counter
        if (fa) return a.f < b.f - 1e-7 * std::fabs(b.f) - 1e-300;
        return a.h < b.h * (1.0 - 1e-7) || (a.h == b.h && a.f < b.f - 1e-7 * std::fabs(b.f));
    };

    // ---- first descent from x0, then variable-neighbourhood restarts (Audet, Bechard & Le Digabel 2008): shake the
    // incumbent by k half-frames along a fresh direction, descend from there with a fresh barrier, keep the outcome only
    // if it beats the incumbent (k back to 1), otherwise widen the shake ----
    MadsPoint best{};
    bool best_feasible = false;
    if (!descent(x0, 0, best, best_feasible)) { res.stop_reason = 3; failed = true; }
    int k = 1, idle = 0;
    while (!failed && res.evaluations < o.max_bb_eval && idle < o.vns_max_idle) {
        double u[n], umax = 0, start[n];
        halton_dir(u);
        for (int i = 0; i < n; ++i) umax = std::max(umax, std::fabs(u[i]));
        for (int i = 0; i < n; ++i) start[i] = best.x[i] + (double)k * o.vns_amplitude * frame0[i] * u[i] / umax;
        MadsPoint loc{};
        bool loc_feasible = false;
        if (!descent(start, o.vns_mesh_index, loc, loc_feasible)) { res.stop_reason = 3; failed = true; break; }
        ++res.restarts;
        if (better(loc, loc_feasible, best, best_feasible)) { best = loc; best_feasible = loc_feasible; k = 1; idle = 0; }
        else { k = std::min(k + 1, o.vns_max_k); ++idle; }
    }
    if (failed) return false;
    res.stop_reason = res.evaluations >= o.max_bb_eval ? 2 : 1;
    res.best = best;
    res.feasible = best_feasible ? 1 : 0;
    return true;
}

}  // namespace iba
