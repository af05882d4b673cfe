// TEST INFRASTRUCTURE (oracle/dropin): what a maintainer puts in the place of matching/SlidingWindowSparseEMMatcher.h to
// build `mbgc` over libmbgc_hip.so. The class of that name now is the facade over the C ABI
// (mbgc_amd/host/sw_matcher.h); the Exp variant MultipleGenomeMatchingProcessor::initMatcher constructs for an even
// k1 (MGMP.cpp:170-172) is the same class — the device path implements that variant (k1 a power of two).
#ifndef PGTOOLS_SWSMEMMATCHER_H
#define PGTOOLS_SWSMEMMATCHER_H

#include "TextMatchers.h"
#include <deque>

using namespace PgTools;

#define MBGC_HIP_USE_REFERENCE_TEXTMATCH      // PgTools::TextMatch is the reference's own (matching/TextMatchers.h)
#include "sw_matcher.h"

typedef SlidingWindowSparseEMMatcher SlidingWindowExpSparseEMMatcher;

#endif
