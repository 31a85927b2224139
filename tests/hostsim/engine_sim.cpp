// engine_sim.cpp -- TEST INFRASTRUCTURE ONLY (see prim_sim.hpp).
// Compiles the engine's kernel bodies and host orchestration over the serial
// stand-in primitives so the CPU test-suite can exercise them without a GPU.
// Exports the same C-ABI symbols as libgrlbwt_hip.so from a DIFFERENT library
// (tests/hostsim/_build/libgrlbwt_sim.so) that only tests/ load.
#include <time.h>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>
#include <algorithm>
#include <string>

#include "prim_sim.hpp"

#define GRL_NS grl32
#define GRL_IDX_T uint32_t
#define GRL_IDX_BYTES 4
#include "../../grlbwt_amd/csrc/engine_impl.hpp"
#undef GRL_NS
#undef GRL_IDX_T
#undef GRL_IDX_BYTES

#define GRL_NS grl64
#define GRL_IDX_T uint64_t
#define GRL_IDX_BYTES 8
#include "../../grlbwt_amd/csrc/engine_impl.hpp"
#undef GRL_NS
#undef GRL_IDX_T
#undef GRL_IDX_BYTES

#include "../../grlbwt_amd/csrc/capi_impl.hpp"
