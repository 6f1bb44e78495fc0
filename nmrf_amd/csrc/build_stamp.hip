// Build stamp (python -m nmrf_amd.build passes -DNMRF_BUILD_STAMP="<sha256/16 of every source and header>"): the product library
// and the tools library of one build carry the same stamp; nmrf_amd/_lib.py and tests/conftest.py refuse a tools library whose stamp
// differs from the product's or from the sources in the tree (the stale-library failure mode of round 3).
#include "common.h"
#ifndef NMRF_BUILD_STAMP
#define NMRF_BUILD_STAMP "unstamped"
#endif
extern "C" const char *nmrf_build_stamp(void) { return "abi" NMRF_ABI_STR "-" NMRF_BUILD_STAMP; }
