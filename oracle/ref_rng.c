/*
 * TEST INFRASTRUCTURE ONLY.  Separate TU because the reference's RNG state is
 * `static _Thread_local` inside utils.c (utils.c:60) and utils.h has no include guard.
 */
#include "utils.c"

__attribute__((visibility("default"))) void     ref_set_rng(uint64_t s) { wyhash64_x = s; }
__attribute__((visibility("default"))) uint64_t ref_get_rng(void)       { return wyhash64_x; }
