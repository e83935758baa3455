// The one-wave-per-SIMD variant of the six-lane kernels (engine_coop.hip explains why it is its own translation unit).
#define RB_C6_W1_UNIT
#include "engine_coop.hip"
