/*
 * TEST INFRASTRUCTURE ONLY.  Verbatim-reference harness TU: textually includes the reference's
 * main.c from where it lies (-I$(REF)/src) with its `main` renamed, then appends the harness.
 */
#define main reference_main
#include "main.c"
#undef main
#include "ref_harness_body.h"
