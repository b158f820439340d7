// intfft_version.hip -- the one translation unit that knows which sources the library was built from.
// intfftk_amd/build.py hashes every file of csrc/ plus include/intfft.h (sha256, first 16 hex digits) and passes it as INTFFT_SRC_HASH;
// the string identifies a build across machines and rebuilds (a binary hash would not: code objects are not bit-reproducible), so the
// committed PMC digests (profiles/*_pmc_digest.json: "lib_version") can be tied to the library bench.py is measuring.
#include "../../include/intfft.h"

#ifndef INTFFT_SRC_HASH
#define INTFFT_SRC_HASH "unhashed"
#endif

extern "C" const char *intfft_version(void) { return "intfft-mi355x 0.6 (gfx950) src " INTFFT_SRC_HASH; }
