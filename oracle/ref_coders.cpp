// TEST INFRASTRUCTURE (oracle/): the two leaf coders the backend of include/mbgc_backend.h calls back into — the reference's
// unchanged PPMd7 / LZMA (coders/PpmdCoder.cpp, coders/LzmaCoder.cpp around the 7-zip SDK in coders/lzma/, public domain) —
// as a library of their own: oracle/_ref/libmbgc_coders.so, built by `make -C oracle coders` from the reference's coder sources
// where they lie (CodersLib, LzmaCoder, PpmdCoder, VarLenDNACoder, SymbolsPackingFacility, helper + coders/lzma/*.c), nothing of
// its matcher, encoder or decoder. This is what `mbgc-hip c --backend` is given; it exports the callback of
// include/mbgc_backend.h under the name the tool looks for.
#include <memory>
#include <ostream>
#include <streambuf>
#include <cstring>
#include <cstdint>
#include <omp.h>

#include "utils/helper.h"
#include "coders/CodersLib.h"
#include "coders/LzmaCoder.h"
#include "coders/PpmdCoder.h"

namespace {
struct NullBuf : std::streambuf { int overflow(int c) override { return c; } };
NullBuf g_nullbuf;
std::ostream g_null(&g_nullbuf);
}

// == mbgc_leaf_coder_t (include/mbgc_backend.h)
struct RefLeafCoder { int coder, level, lc, lp, pb, fb, algo, numThreads; uint32_t dictSize, memSize; int order; };

extern "C" int mbgc_leaf_compress(void *, const RefLeafCoder *c, const unsigned char *src, uint64_t n, unsigned char *dest, uint64_t cap, uint64_t *destLen) {
    // one leaf coder call as Compress() makes it (coders/CodersLib.cpp:53-66); may be called from several threads at once
    PgHelpers::devout = &g_null;
    PgHelpers::appout = &g_null;
    std::unique_ptr<CoderProps> props;
    if (c->coder == LZMA_CODER) props.reset(new LzmaCoderProps(c->level, c->dictSize, c->lc, c->lp, c->pb, c->fb, c->algo, c->numThreads));
    else if (c->coder == PPMD7_CODER) props.reset(new PpmdCoderProps(c->memSize, c->order));
    else return -1;
    size_t len = 0;
    unsigned char *out = Compress(len, src, n, props.get(), 1, &g_null);
    if (len > cap) { delete[] out; return -2; }
    memcpy(dest, out, len);
    delete[] out;
    *destLen = len;
    return 0;
}
