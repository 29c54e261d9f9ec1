// evalh_types.hpp -- host-visible layout of a compiled quotient-numerator program (shared by the
// C-ABI translation unit, which builds it, and evalh.cuh, which runs it).
#pragma once
#include <cstdint>

#include "fp.cuh"

#define EVH_THREADS 128
#define EVH_LDS_BYTES (60 * 1024)
#define EVH_SLOT_BYTES (36 * EVH_THREADS)
#define EVH_MAX_LDS_SLOTS (EVH_LDS_BYTES / EVH_SLOT_BYTES)      // 13

enum : uint32_t { EVS_SCALAR = 0, EVS_SLOT_LDS = 1, EVS_SLOT_HBM = 2, EVS_FIXED = 3, EVS_ADVICE = 4, EVS_INSTANCE = 5, EVS_PREVIOUS = 6 };

struct DevSrc {
    uint32_t kind;
    uint32_t index;      // scalar-table index | slot | column index
    int32_t rot;    // column rotation (rows of the ORIGINAL domain)
};
struct DevCalc {
    uint32_t op;
    uint32_t target_kind, target_slot;
    uint32_t parts_begin, parts_len;
    DevSrc a, b;
};

struct dehalo_graph {
    int field;
    uint32_t num_calcs, num_parts, num_constants, lds_slots, hbm_slots;
    uint32_t max_fixed, max_advice, max_instance, max_challenge;   // highest index referenced + 1
    bool uses_previous;
    DevCalc* d_calcs;
    DevSrc* d_parts;
    fe* d_constants;        // internal packed form
    DevSrc result;          // where the last calculation's value lives
};

