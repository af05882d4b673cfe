#include "simple_sequence_matcher.h"

#include <cstdio>
#include <cstdlib>

#include "../../include/mbgc_copmem.h"

namespace PgTools {

void SimpleSequenceMatcher::rcMatchSequence(std::string &sequence, std::string &rcMapOff, std::string &rcMapLen, size_t targetMatchLength,
                                            uint32_t minMatchLength, int device) {
    mbgc_copmem_t *h = nullptr;
    // the reference prints its message and exits (CopMEMMatcher.cpp:76-79,:118-121,:522-525); so does this layer
    if (mbgc_copmem_create(&h, device) != 0) { fprintf(stderr, "%s\n\n", mbgc_copmem_last_error()); exit(EXIT_FAILURE); }
    uint64_t newLen = 0, nOff = 0, nLen = 0;
    const uint8_t *off = nullptr, *len = nullptr;
    const int rc = mbgc_copmem_rc_match_sequence(h, (uint8_t *) &sequence[0], sequence.size(), (uint32_t) targetMatchLength, minMatchLength, &newLen,
                                                 &off, &nOff, &len, &nLen, nullptr);
    if (rc == -101 || rc == -103) {
        // The literal stream is too long for the device pass (its working set is 8-10 x the stream; sample numbers are 32-bit),
        // found out at the very end of an encode: the pass is an optional redundancy removal — the stream stays as it is and the
        // two maps stay empty, which is a valid archive (no reverse-complement match recorded), instead of losing the whole run.
        fprintf(stderr, "WARNING: reverse-complement pass over the literals skipped (%s)\n", mbgc_copmem_last_error());
        rcMapOff.clear(); rcMapLen.clear();
        mbgc_copmem_destroy(h);
        return;
    }
    if (rc != 0) {
        fprintf(stderr, "%s\n\n", mbgc_copmem_last_error());
        exit(EXIT_FAILURE);
    }
    sequence.resize(newLen);
    rcMapOff.assign((const char *) off, nOff);
    rcMapLen.assign((const char *) len, nLen);
    mbgc_copmem_destroy(h);
}

}  // namespace PgTools
