// PgTools::SimpleSequenceMatcher::rcMatchSequence with the reference's signature (matching/SimpleSequenceMatcher.h:38-39,
// .cpp:165-176) over the C ABI of include/mbgc_copmem.h: the `-m3` reverse-complement pass over the literal stream, index
// build and query scan on the device (mbgc_amd/csrc/copmem.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace PgTools {

class SimpleSequenceMatcher {
public:
    // sequence is rewritten in place (matched parts cut out, RC_MATCH_MARK left behind); rcMapOff / rcMapLen receive the maps
    static void rcMatchSequence(std::string &sequence, std::string &rcMapOff, std::string &rcMapLen, size_t targetMatchLength,
                                uint32_t minMatchLength = UINT32_MAX, int device = 0);
};

}  // namespace PgTools
