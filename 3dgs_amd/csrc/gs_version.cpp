// gs_version.cpp -- what this build of libgsplat_hip.so was made from.
#ifndef GS_BUILD_FLAGS
#define GS_BUILD_FLAGS ""
#endif
extern "C" {
// sha256 (first 16 hex digits) over Makefile, *.h and *.hip of this directory as they were when the library was linked
const char *gsplat_source_hash(void) {
  return
#include "gs_source_hash.inc"
      ;
}
// the EXTRA compiler flags of a diagnostic / experiment build (GS_STAMP, GS_ABLATE, ...); empty for the product build
const char *gsplat_build_flags(void) { return GS_BUILD_FLAGS; }
}
