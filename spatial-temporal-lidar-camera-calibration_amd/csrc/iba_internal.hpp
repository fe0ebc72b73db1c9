// Library-internal interface between the single-device entry points (iba_capi.hip) and the multi-GPU group (iba_group.hip):
// the host-side candidate block is computed ONCE per call (Sim3Exp, SE3Exp(-x) and their duals: ~0.7 us per candidate) and
// handed to every device's launch chain, instead of once per device.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/iba_mi355x.h"
#include "iba_types.hpp"

namespace iba {

enum EvalKind { kEvalCost = 0, kEvalNormal = 1, kEvalFull = 2, kEvalFactors = 3 };

// Cand block of B candidates x (7 doubles each): g2o_tools.h:105-140, 149-183 on the host
void make_cands_host(const double* x, int B, Cand* out, bool jets = true);   // jets = false: the values only (cost evaluations)
// launch chain of one chunk (B <= chain_capacity(h)) on `st` from a ready candidate block (host memory, copied into the handle's
// pinned ring before the call returns); no synchronisation
// jets_ready != nullptr (kEvalNormal / kEvalFull): the derivative half of host_cands is still being computed by the caller; the chain
// starts on the values and copies the block again, before the factor kernel, once *jets_ready is set
iba_status eval_partial_cands(iba_handle* h, const Cand* host_cands, int B, EvalKind kind, double* d_partials, hipStream_t st, const std::atomic<int>* jets_ready = nullptr);
void make_cands_jets_host(const double* x, int B, Cand* out);   // the derivative half alone
// iba_build_problem from a ready candidate (synchronises the handle's stream: the frozen counts are read back)
iba_status build_problem_cands(iba_handle* h, const Cand* host_cand);
// work buffers for batches of up to B candidates, so that no evaluation allocates
iba_status reserve_batch(iba_handle* h, int B);
// getenv(name) when IBA_DEBUG_ENV=1 is set, nullptr otherwise: every environment override of the library is a debug aid (include/iba_mi355x_debug.h)
const char* debug_env(const char* name);
// candidates one launch chain takes on this handle (iba_create_options.max_chain_batch; IBA_MAX_BATCH while the planes are refitted per evaluation)
int chain_capacity(const iba_handle* h);

}  // namespace iba
