// n1k_jit.h — plan-specialised kernels compiled at run time.
//
// scan_spec_kernel (n1k_spec.h) is a hand-written template over the SHAPE of a plan.  A registry of shapes is
// instantiated ahead of time; for any other bounded shape the same template is instantiated here through hiprtc
// (in-process compiler, ~1-2 s, cached per shape for the life of the process) — the query-compiler step the
// reference would attach to a prepared statement (plan/prepared.go).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "n1k_types.h"

namespace n1k {

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t wide = nullptr;    // scan_spec_body<Spec, 2, 512, true>
    hipFunction_t narrow = nullptr;  // scan_spec_body<Spec, 4, 512, false>
    hipFunction_t rec_wide = nullptr, rec_narrow = nullptr;  // scan_spec_records_body<Spec, 2 / 4, 512, true / false> (partitioned GROUP BY front end)
    hipFunction_t part_wide = nullptr, part_narrow = nullptr, part_wide256 = nullptr;  // scan_spec_partition_body<...> (SpecSig::mode 1: the row exchange)
    bool failed = false;
    std::string log;
};

// Source text of the translation unit for one shape (exposed for tests / debugging).
std::string jit_source(const SpecSig& sig);
// Compile (or fetch from the cache) the kernels of a shape.  Never throws; on failure ->failed is set and ->log
// holds the compiler output.
const JitKernel* jit_get(const SpecSig& sig);
// Compile only (no GPU needed): returns true when the shape builds for gfx950.
bool jit_compile_check(const SpecSig& sig, std::string* log);
hipError_t jit_launch(const JitKernel* k, const Program& P, const FastArgs& F, const GlobalTable& G,
                      unsigned long long* ngroups, uint32_t grid, bool wide, const WordLogArgs& L, uint32_t ndistinct, hipStream_t st);

hipError_t jit_launch_records(const JitKernel* k, const Program& P, const FastArgs& F, uint32_t grid, bool wide, const WordLogArgs& L,
                              hipStream_t st);

hipError_t jit_launch_partition(const JitKernel* k, const Program& P, const FastArgs& F, const PartArgs& A, uint32_t grid, bool wide,
                                uint32_t block, hipStream_t st);

}  // namespace n1k
