// engine_hip.hip -- the one translation unit of libgrlbwt_hip.so (gfx950 only).
// prim_hip.hpp: hand-written device primitives; engine_impl.hpp: the engine,
// instantiated for 32-bit and 64-bit positions/lengths; capi_impl.hpp: C-ABI.
#include <time.h>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>
#include <algorithm>
#include <string>

#include "prim_hip.hpp"

#define GRL_NS grl32
#define GRL_IDX_T uint32_t
#define GRL_IDX_BYTES 4
#include "engine_impl.hpp"
#undef GRL_NS
#undef GRL_IDX_T
#undef GRL_IDX_BYTES

#define GRL_NS grl64
#define GRL_IDX_T uint64_t
#define GRL_IDX_BYTES 8
#include "engine_impl.hpp"
#undef GRL_NS
#undef GRL_IDX_T
#undef GRL_IDX_BYTES

#include "capi_impl.hpp"
